"""Where the cloud preparation of a 1M<->1M registration goes: host time of the two set calls, of the finalize
(bounding-box round trip + enqueue of the grid ladder / Morton sort), device drain, then the align."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "4")
import numpy as np, torch
from libwave_amd import capi, synth
ref, tgt, T_gt = synth.pair(1000000, seed=42)
a, b = torch.from_numpy(ref).cuda(), torch.from_numpy(tgt).cuda()
ctx = capi.Context(0)
def reg():
    ctx.set_source(a); ctx.set_target(b)
    return ctx.icp_align(max_corr=3.0, force_iterations=50, nn_method=capi.WM_NN_GRID, carry_state=0)
for _ in range(4): reg()
rows = []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); ctx.set_source(a); ctx.set_target(b)
    t1 = time.perf_counter(); ctx.nn_search(np.eye(4), 3.0, capi.WM_NN_GRID)   # finalize + ladder + one search, waits
    t2 = time.perf_counter()
    rows.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3))
print("set calls %.3f ms; finalize + ladder + first search (waited) %.3f ms" % tuple(np.median(np.array(rows), axis=0)))
ts = []
for _ in range(10):
    t0 = time.perf_counter(); r = reg(); ts.append((time.perf_counter() - t0) * 1e3)
print("registration %.3f ms, align_ms (events around the loop) %.3f -> outside the loop %.3f ms" % (np.median(ts), r["align_ms"], np.median(ts) - r["align_ms"]))
