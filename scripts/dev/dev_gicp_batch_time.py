"""Developer: 256 distinct 20k-point GICP pairs per launch, both objectives -- kernel ms, pairs/s."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ.setdefault("OMP_NUM_THREADS", "4")
import numpy as np, torch
from libwave_amd import capi, synth
n, B = 20000, 256
base = [synth.pair(n, seed=300 + k, mode="resample")[:2] for k in range(B)]
dev = [(torch.from_numpy(r).cuda(), torch.from_numpy(t).cuda()) for r, t in base]
ctx = capi.Context(0)
for name, obj in (("statistics", 0), ("pcl_sums", 1), ("statistics", 0), ("pcl_sums", 1)):
    walls, kern = [], []
    for rep in range(5):
        t0 = time.perf_counter()
        got = ctx.gicp_batch_match(dev, objective=obj)
        walls.append((time.perf_counter() - t0) * 1e3)
        kern.append(got[0]["kernel_ms"])
    print("%s: wall %s kernel %s -> %.0f pairs/s; all ok %s; outer %s evals %s" % (
        name, [round(w, 1) for w in walls], [round(k, 1) for k in kern], B / min(walls) * 1e3, all(g["rc"] == 0 for g in got),
        [g["iterations"] for g in got[:6]], [g["evaluations"] for g in got[:6]]))
