import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from libwave_amd import capi, synth
ctx = capi.Context(0)
pairs = [synth.pair(n, seed=7 + k, mode="resample")[:2] for k, n in enumerate((3000, 9000, 15000))]
for kw in (dict(max_corr=1e-3), dict(max_corr=1e3, max_iter=5), dict(max_corr=3.0, max_iter=1), dict(max_corr=3.0, force_iterations=3, mode=capi.WM_ICP_GN6),
           dict(max_corr=0.5, res=0.3, multiscale_steps=2), dict(max_corr=3.0, res=5.0, multiscale_steps=1), dict(max_corr=3.0, res=1e-4, multiscale_steps=0)):
    t0 = time.perf_counter()
    got = ctx.icp_batch_match(pairs, with_info=True, **kw)
    one = []
    for r, t in pairs:
        c = capi.Context(0)
        k2 = dict(kw); res = k2.pop("res", -1.0); ms = k2.pop("multiscale_steps", 0)
        one.append(c.icp_match(r, t, res=res, multiscale_steps=ms, carry_state=1, **k2)); c.close()
    print(kw, "%.1f ms" % ((time.perf_counter() - t0) * 1e3), [(g["rc"], g["iterations"], g["n_corr"]) for g in got], [(o["rc"], o["iterations"], o["n_corr"]) for o in one], flush=True)
