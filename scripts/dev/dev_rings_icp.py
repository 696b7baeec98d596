"""ICP 1M<->1M on the 64-ring lidar sampling of the scene (non-uniform density) against the uniform one."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import torch
from libwave_amd import capi, synth
ctx = capi.Context(0)
for pattern in ("uniform", "rings"):
    kw = {} if pattern == "uniform" else {"pattern": "rings"}
    ref, tgt, T_gt = synth.pair(1_000_000, seed=42, **kw)
    d_ref, d_tgt = torch.from_numpy(ref).cuda(), torch.from_numpy(tgt).cuda()
    def step(profile=0):
        ctx.set_source(d_ref); ctx.set_target(d_tgt)
        return ctx.icp_align(max_corr=3.0, force_iterations=50, nn_method=capi.WM_NN_GRID, profile=profile, carry_state=0)
    for _ in range(3): step()
    ts = []
    for _ in range(8):
        t0 = time.perf_counter(); r = step(); ts.append(time.perf_counter() - t0)
    r1 = step(1)
    it = ctx.iteration_times() * 1e3
    print("%s: %.3f ms/registration, nn/launch %.1f us, err_t %.2e, n_corr %d, grid cell %.3f; nn us by iteration %s" % (
        pattern, np.median(ts) * 1e3, r1["nn_ms"] / 50 * 1e3, np.linalg.norm(r["T"][:3, 3] - T_gt[:3, 3]), r["n_corr"], r["grid_cell"],
        " ".join("%d:%.0f" % (k, it[k]) for k in range(len(it)))), flush=True)
    print("   cert launches", r1.get("cert_launches"), "prep+rest = %.3f ms" % (np.median(ts) * 1e3 - r1["nn_ms"]), flush=True)
