"""Soak of the batched GICP / NDT paths: hundreds of pairs of random sizes (tiny ones, non-finite points, shifted clouds),
several calls; every sampled item against the one-pair path."""
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np

from libwave_amd import capi, synth

rng = np.random.default_rng(5)
ctx = capi.Context(0)
base = [synth.pair(30000, seed=40 + k, mode="resample") for k in range(6)]


def make(k):
    r, t, _ = base[k % 6]
    n = int(rng.choice([3, 9, 11, 40, 300, 2000, 7000, 15000, 30000]))
    a = rng.permutation(30000)[:n]
    b = rng.permutation(30000)[: max(3, int(n * rng.uniform(0.5, 1.0)))]
    rr, tt = r[a].copy(), t[b].copy()
    if rng.uniform() < 0.2 and n > 20:
        rr[rng.integers(0, n, n // 10)] = np.nan
    if rng.uniform() < 0.1:
        tt = tt + np.float32(rng.choice([0.5, 3.0, 40.0]))
    return rr, tt


bad = 0
for call in range(4):
    pairs = [make(k) for k in range(int(rng.integers(100, 420)))]
    t0 = time.perf_counter()
    g = ctx.gicp_batch_match(pairs)
    t1 = time.perf_counter()
    d = ctx.ndt_batch_match(pairs, res=float(rng.choice([1.0, 2.0, 5.0])))
    t2 = time.perf_counter()
    print("call %d: %d pairs, gicp %.1f ms, ndt %.1f ms; gicp rc histogram %s, ndt %s" % (
        call, len(pairs), (t1 - t0) * 1e3, (t2 - t1) * 1e3, np.bincount([x["rc"] + 4 for x in g], minlength=8).tolist(),
        np.bincount([x["rc"] + 4 for x in d], minlength=8).tolist()), flush=True)
    for k in rng.integers(0, len(pairs), 12):
        o = ctx.gicp_match(*pairs[k]) if len(pairs[k][0]) and len(pairs[k][1]) else None
        if o is not None and (o["rc"] != g[k]["rc"] or (o["T"] is not None and np.abs(o["T"] - g[k]["T"]).max() > 1e-4)):
            bad += 1
            print("  gicp item %d differs: rc %d / %d, sizes %d %d, iterations %d / %d" % (k, o["rc"], g[k]["rc"], len(pairs[k][0]), len(pairs[k][1]), o["iterations"], g[k]["iterations"]))
print("differing sampled items:", bad)
