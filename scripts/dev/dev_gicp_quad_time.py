"""Developer: the two GICP objectives on the 500k pair of BASELINE configs[2] -- ms per registration, iterations."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ.setdefault("OMP_NUM_THREADS", "4")
import numpy as np, torch
from libwave_amd import capi, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
ref, tgt, T_gt = synth.pair(n, seed=42)
d_ref, d_tgt = torch.from_numpy(ref).cuda(), torch.from_numpy(tgt).cuda()
ctx = capi.Context(0)
for name, obj in (("statistics", 0), ("pcl_sums", 1)):
    ts = []
    for rep in range(6):
        t0 = time.perf_counter()
        ctx.set_source(d_ref); ctx.set_target(d_tgt)
        r = ctx.gicp_align(objective=obj)
        ts.append((time.perf_counter() - t0) * 1e3)
    print("%s: %.3f ms (min %.3f) outer %d inner %d evals %d rc %d err_t %.2e" % (
        name, float(np.median(ts[1:])), min(ts), r["iterations"], r["inner_total"], r["evaluations"], r["rc"],
        np.linalg.norm(r["T"][:3, 3] - T_gt[:3, 3])))
