"""Batched small NDT (wm_ndt_batch_match) against the one-pair path on the same pairs: agreement and throughput."""
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np

from libwave_amd import capi, synth

n = int(os.environ.get("NDT_POINTS", "20000"))
B = int(os.environ.get("NDT_PAIRS", "8"))
res = float(os.environ.get("NDT_RES", "1.0"))
pairs, gts = [], []
for k in range(B):
    ref, tgt, T_gt = synth.pair(n, seed=100 + (k % 16), mode="resample")
    pairs.append((ref, tgt))
    gts.append(T_gt)
ctx = capi.Context(0)
one = []
t0 = time.perf_counter()
for ref, tgt in pairs[: min(B, 8)]:
    ctx.set_source(ref)
    ctx.set_target(tgt)
    one.append(ctx.ndt_align(res=res))
t_one = (time.perf_counter() - t0) / max(len(one), 1)
for rep in range(3):
    t0 = time.perf_counter()
    got = ctx.ndt_batch_match(pairs, res=res)
    t_b = time.perf_counter() - t0
    print("batch of %d pairs of %d points, res %.2f: %.2f ms (kernel %.2f ms) = %.0f pairs/s | one-pair path %.2f ms each" % (
        B, n, res, t_b * 1e3, got[0]["kernel_ms"], B / t_b, t_one * 1e3), flush=True)
for k, r in enumerate(one):
    g = got[k]
    dT = np.abs(r["T"] - g["T"]).max() if r["T"] is not None and g["T"] is not None else float("nan")
    print("pair %d: rc %d / %d, iterations %d / %d, passes %d / %d, voxels %d / %d, max |dT| %.3e, |t - t_gt| %.2e / %.2e, score %.9g / %.9g" % (
        k, r["rc"], g["rc"], r["iterations"], g["iterations"], r["evaluations"], g["evaluations"], r["n_voxels"], g["n_voxels"], dT,
        np.linalg.norm(r["T"][:3, 3] - gts[k][:3, 3]) if r["T"] is not None else -1,
        np.linalg.norm(g["T"][:3, 3] - gts[k][:3, 3]) if g["T"] is not None else -1, r["score"], g["score"]))
