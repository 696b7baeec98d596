"""Developer probe: GICPMatcher::match() and NDTMatcher::match() with the reference's DEFAULT
parameters (GICP: res = 0.1 voxel filter of both clouds, corr_rand 10; NDT: res = 5 m voxels,
step_size 3) on 500k / 1M-point device clouds."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "4")
import numpy as np, torch
from libwave_amd import capi, synth
ctx = capi.Context(0)


def timed(fn, reps=5):
    fn(); fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts)), r


for n in (500_000, 1_000_000):
    ref, tgt, T_gt = synth.pair(n, seed=42)
    d_ref, d_tgt = torch.from_numpy(ref).cuda(), torch.from_numpy(tgt).cuda()
    ms, r = timed(lambda: ctx.gicp_match(d_ref, d_tgt, res=0.1))
    err = np.linalg.norm(r["T"][:3, 3] - T_gt[:3, 3]) if r["T"] is not None else -1
    print("GICP default (res=0.1) %d pts: %.2f ms/match rc=%d outer=%s evals=%s err=%.4f" % (
        n, ms, r["rc"], r.get("iterations"), r.get("evaluations"), err))

    def ndt():
        ctx.set_source(d_ref); ctx.set_target(d_tgt)
        return ctx.ndt_align(res=5.0, step_size=3.0)
    ms, r = timed(ndt)
    err = np.linalg.norm(r["T"][:3, 3] - T_gt[:3, 3]) if r["T"] is not None else -1
    print("NDT default (res=5) %d pts: %.2f ms/match rc=%d iters=%d passes=%d err=%.4f" % (
        n, ms, r["rc"], r["iterations"], r["evaluations"], err))
