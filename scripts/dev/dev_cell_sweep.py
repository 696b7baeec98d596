"""Level-0 cell size of the search grid against registration time (1M<->1M, 50 forced iterations)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import torch
from libwave_amd import capi, synth
pattern = os.environ.get("PATTERN", "")  # "rings": the 64-ring scan
ref, tgt, T_gt = synth.pair(1_000_000, seed=42, pattern=pattern) if pattern else synth.pair(1_000_000, seed=42)
d_ref, d_tgt = torch.from_numpy(ref).cuda(), torch.from_numpy(tgt).cuda()
for h in [float(x) for x in os.environ.get("CELLS", "0.0,0.09,0.11,0.13,0.15,0.183,0.22,0.27").split(",")]:
    ctx = capi.Context(0)
    if h > 0:
        ctx.set_grid_cell(h)
    def step(profile=0):
        ctx.set_source(d_ref); ctx.set_target(d_tgt)
        return ctx.icp_align(max_corr=3.0, force_iterations=50, nn_method=capi.WM_NN_GRID, profile=profile, carry_state=0)
    for _ in range(3): step()
    ts = []
    for _ in range(8):
        t0 = time.perf_counter(); r = step(); ts.append(time.perf_counter() - t0)
    r1 = step(1)
    it = ctx.iteration_times() * 1e3
    print("h=%.3f (used %.3f): %.3f ms/registration, nn/launch %.1f us; by iteration %s" % (
        h, r["grid_cell"], np.median(ts) * 1e3, r1["nn_ms"] / 50 * 1e3, " ".join("%d:%.0f" % (k, it[k]) for k in (0, 1, 4, 8, 16, 49))), flush=True)
    ctx.close()
