"""Throughput of wm_icp_batch_match on one MI355X: B pairs of n points per launch, host clouds.
usage: dev_batch.py [n=10000] [B=256 ...]"""
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import torch

from libwave_amd import capi, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
Bs = [int(a) for a in sys.argv[2:]] or [64, 256, 512]
base = [synth.pair(n, seed=100 + k, mode="resample")[:2] for k in range(8)]
dev = [(torch.from_numpy(r).cuda(), torch.from_numpy(t).cuda()) for r, t in base]
ctx = capi.Context(0)
for B in Bs:
    for label, src in (("host", base), ("device", dev)):
        pairs = [src[k % len(src)] for k in range(B)]
        for with_info in (False, True):
            ctx.icp_batch_match(pairs, with_info=with_info, max_corr=3.0, max_iter=100)
            ts = []
            for _ in range(5):
                t0 = time.perf_counter()
                got = ctx.icp_batch_match(pairs, with_info=with_info, max_corr=3.0, max_iter=100)
                ts.append(time.perf_counter() - t0)
            t = float(np.median(ts))
            its = [g["iterations"] for g in got[:8]]
            print("n=%d B=%d %s info=%d: %.2f ms/batch  %.0f registrations/s  kernel %.2f ms  iterations %s" % (
                n, B, label, with_info, t * 1e3, B / t, got[0]["align_ms"], its), flush=True)
