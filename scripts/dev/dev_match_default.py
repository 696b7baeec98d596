"""Developer probe: ICPMatcher::match() with the reference's DEFAULT parameters (res = 0.1,
multiscale_steps = 3, max_corr = 3, max_iter = 100, t_eps = 1e-8, fit_eps = 1e-2) on 1M-point
clouds -- the path a libwave user gets without touching the config."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from libwave_amd import capi, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
ref, tgt, T_gt = synth.pair(n, seed=42)
ctx = capi.Context(0)
d_ref, d_tgt = torch.from_numpy(ref).cuda(), torch.from_numpy(tgt).cuda()
for name, (a, b) in {"host": (ref, tgt), "device": (d_ref, d_tgt)}.items():
    for ms in (0, 3):
        for _ in range(2):
            r = ctx.icp_match(a, b, res=0.1, multiscale_steps=ms, max_corr=3.0, max_iter=100, t_eps=1e-8, fit_eps=1e-2)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            r = ctx.icp_match(a, b, res=0.1, multiscale_steps=ms, max_corr=3.0, max_iter=100, t_eps=1e-8, fit_eps=1e-2)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5 * 1e3
        err = np.linalg.norm(r["T"][:3, 3] - T_gt[:3, 3]) if r["T"] is not None else -1
        print("%s clouds, res=0.1, multiscale_steps=%d: %.2f ms/match, rc=%d iters=%d err=%.4f" % (name, ms, dt, r["rc"], r["iterations"], err))
