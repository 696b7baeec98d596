import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ.setdefault("OMP_NUM_THREADS","4"); os.environ.setdefault("OPENBLAS_NUM_THREADS","4")
import numpy as np, torch
from libwave_amd import capi, synth
ref, tgt, T_gt = synth.pair(1_000_000, seed=42)
ctx = capi.Context(0)
d_ref, d_tgt = torch.from_numpy(ref).cuda(), torch.from_numpy(tgt).cuda()
for rep in range(3):
    ctx.set_source(d_ref); ctx.set_target(d_tgt)
    r = ctx.icp_align(max_corr=3.0, force_iterations=50, nn_method=capi.WM_NN_GRID, profile=1, carry_state=0)
t = ctx.iteration_times() if hasattr(ctx, "iteration_times") else None
print([round(x*1e3) for x in t] if t is not None else r.keys())
