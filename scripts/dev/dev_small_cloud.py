import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ.setdefault("OMP_NUM_THREADS","4"); os.environ.setdefault("OPENBLAS_NUM_THREADS","4")
import numpy as np, torch
from libwave_amd import capi, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
ref, tgt, T_gt = synth.pair(n, seed=42)
ctx = capi.Context(0)
d_ref, d_tgt = torch.from_numpy(ref).cuda(), torch.from_numpy(tgt).cuda()
def reg():
    ctx.set_source(d_ref); ctx.set_target(d_tgt)
    return ctx.icp_align(max_corr=3.0, max_iter=100, t_eps=1e-8, fit_eps=1e-2, carry_state=0)
for _ in range(3): r = reg()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): r = reg()
torch.cuda.synchronize(); print("n=%d: %.3f ms/registration, iters=%d" % (n, (time.perf_counter()-t0)/50*1e3, r["iterations"]))
