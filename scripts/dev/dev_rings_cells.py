"""64-ring 1M pair: level-0 cell size x ordering of crowded cells (WM_TUNE_SORT_HEAVY), per-iteration search times."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import torch
from libwave_amd import capi, synth
pattern = os.environ.get("RC_PATTERN", "rings")
kw = {} if pattern == "uniform" else {"pattern": "rings"}
ref, tgt, T_gt = synth.pair(1_000_000, seed=42, **kw)
d_ref, d_tgt = torch.from_numpy(ref).cuda(), torch.from_numpy(tgt).cuda()
for sort in (0, 1):
    os.environ["WM_TUNE_SORT_HEAVY"] = str(sort)
    for h in [float(v) for v in os.environ.get("RC_CELLS", "0,0.13,0.183,0.25").split(",")]:
        ctx = capi.Context(0)
        if h > 0:
            ctx.set_grid_cell(h)
        def step(profile=0):
            ctx.set_source(d_ref); ctx.set_target(d_tgt)
            return ctx.icp_align(max_corr=3.0, force_iterations=50, nn_method=capi.WM_NN_GRID, profile=profile, carry_state=0)
        for _ in range(3): step()
        ts = []
        for _ in range(6):
            t0 = time.perf_counter(); r = step(); ts.append(time.perf_counter() - t0)
        r1 = step(1)
        it = ctx.iteration_times() * 1e3
        print("%s sort=%d h=%.3f (used %.3f): %.3f ms/registration (align %.3f), nn/launch %.1f us, cert launches %d; by iteration %s" % (
            pattern, sort, h, r["grid_cell"], np.median(ts) * 1e3, r["align_ms"], r1["nn_ms"] / 50 * 1e3, r["cert_launches"],
            " ".join("%d:%.0f" % (k, it[k]) for k in (0, 1, 2, 4, 8, 16, 24, 30, 40, 49))), flush=True)
        ctx.close()
