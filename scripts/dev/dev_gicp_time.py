"""GICP 500k<->500k (BASELINE configs[2]): ms per registration, evaluations, result -- with the
objective evaluated by a kernel launch each (served = 0) and by the resident evaluator (served = 1)."""
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import torch

from libwave_amd import capi, synth

n = int(os.environ.get("GICP_POINTS", "500000"))
ref, tgt, T_gt = synth.pair(n, seed=42)
d_ref, d_tgt = torch.from_numpy(ref).cuda(), torch.from_numpy(tgt).cuda()
res = {}
for posted in (0, 2, 1):
    ctx = capi.Context(0)
    ctx.set_option("gicp_served", posted)

    def run():
        ctx.set_source(d_ref)
        ctx.set_target(d_tgt)
        return ctx.gicp_align()
    run()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        r = run()
        ts.append((time.perf_counter() - t0) * 1e3)
    print("served %d: %.3f ms/registration (min %.3f), %d outer / %d inner iterations, %d evaluations (%d served), f %.17g, |t - t_gt| %.2e" % (
        posted, np.median(ts), min(ts), r["iterations"], r["inner_total"], r["evaluations"], r["served_evaluations"], r["f"],
        np.linalg.norm(r["T"][:3, 3] - T_gt[:3, 3])), flush=True)
    res.setdefault(posted, r)
    ctx.close()
for k in (1, 2):
    print("served=%d vs launched: identical transform:" % k, bool(np.array_equal(res[0]["T"], res[k]["T"])), "| identical objective:", res[0]["f"] == res[k]["f"],
          "| same evaluations:", res[0]["evaluations"] == res[k]["evaluations"])
