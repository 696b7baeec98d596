"""Developer: the NDT derivative passes' neighbour-list certificate on the 2M ring pair of BASELINE configs[3]:
ms per registration with / without, identical results, and (WM_TRACE=1) how many points took their list per pass."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ.setdefault("OMP_NUM_THREADS", "4")
import numpy as np, torch
from libwave_amd import capi, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
res = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
ref, tgt, T_gt = synth.pair(n, seed=42, pattern="rings")
d_ref, d_tgt = torch.from_numpy(ref).cuda(), torch.from_numpy(tgt).cuda()
out = {}
for cache in (1, 0, 1, 0):
    ctx = capi.Context(0)
    ctx.set_option("ndt_cache", cache)
    ts = []
    for rep in range(5):
        t0 = time.perf_counter()
        ctx.set_source(d_ref); ctx.set_target(d_tgt)
        r = ctx.ndt_align(res=res)
        ts.append((time.perf_counter() - t0) * 1e3)
    out[cache] = r
    print("cache %d: %.3f ms (min %.3f) iterations %d passes %d rc %d score %.12g err_t %.2e" % (
        cache, float(np.median(ts[1:])), min(ts), r["iterations"], r["evaluations"], r["rc"], r["score"],
        np.linalg.norm(r["T"][:3, 3] - T_gt[:3, 3])))
    ctx.close()
print("identical:", np.array_equal(out[0]["T"], out[1]["T"]), out[0]["score"] == out[1]["score"], out[0]["evaluations"] == out[1]["evaluations"])
