import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "4")
import numpy as np, torch
from libwave_amd import capi, synth
ref, tgt, T_gt = synth.pair(1000000, seed=42)
pr, pt = torch.from_numpy(ref).pin_memory().numpy(), torch.from_numpy(tgt).pin_memory().numpy()
for cfg in ("1", "0", "1", "0"):
    os.environ["WM_TUNE_EARLY_SOURCE"] = cfg
    ctx = capi.Context(0)
    def step():
        ctx.set_source(pr); ctx.set_target(pt)
        return ctx.icp_align(max_corr=3.0, force_iterations=50, nn_method=capi.WM_NN_GRID, carry_state=0)
    for _ in range(3): r = step()
    ts = []
    for _ in range(10):
        t0 = time.perf_counter(); r = step(); ts.append((time.perf_counter() - t0) * 1e3)
    print("early_source", cfg, "median %.3f ms min %.3f" % (np.median(ts), min(ts)), "T err", float(np.linalg.norm(r["T"][:3, 3] - T_gt[:3, 3])), flush=True)
    ctx.close()
