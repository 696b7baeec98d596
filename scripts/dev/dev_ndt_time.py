"""NDT 2M-point 64-ring pair, 0.5 m voxels (BASELINE configs[3]): ms per registration by WM_TUNE_NDT_BLOCKS."""
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import torch

from libwave_amd import capi, synth

n = int(os.environ.get("NDT_POINTS", "2000000"))
ref, tgt, T_gt = synth.pair(n, seed=42, pattern="rings")
d_ref, d_tgt = torch.from_numpy(ref).cuda(), torch.from_numpy(tgt).cuda()
for blocks, spec in ((1024, 0), (1024, 1)):
    os.environ["WM_TUNE_NDT_BLOCKS"] = str(blocks)
    os.environ["WM_TUNE_NDT_SPEC_HESSIAN"] = str(spec)
    ctx = capi.Context(0)

    def run():
        ctx.set_source(d_ref)
        ctx.set_target(d_tgt)
        return ctx.ndt_align(res=0.5)
    run()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        r = run()
        ts.append((time.perf_counter() - t0) * 1e3)
    print(("blocks %d spec_hessian " + str(spec) + ": %.3f ms/registration (min %.3f), %d iterations, %d passes, |t - t_gt| %.2e") % (
        blocks, np.median(ts), min(ts), r["iterations"], r["evaluations"], np.linalg.norm(r["T"][:3, 3] - T_gt[:3, 3])), flush=True)
    print('   T[:3,3] =', r['T'][:3, 3].tolist(), 'score', r.get('score'))
    ctx.close()
