"""Where a batched small registration spends its time: kernel ms at 1 / 6 / 11 / 21 forced iterations."""
import os
import sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import torch
from libwave_amd import capi, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
base = [synth.pair(n, seed=100 + k, mode="resample")[:2] for k in range(8)]
dev = [(torch.from_numpy(r).cuda(), torch.from_numpy(t).cuda()) for r, t in base]
pairs = [dev[k % 8] for k in range(B)]
ctx = capi.Context(0)
for it in (1, 2, 6, 11, 21, 41):
    for info in (False, True):
        ms = []
        for _ in range(4):
            got = ctx.icp_batch_match(pairs, with_info=info, max_corr=3.0, force_iterations=it)
            ms.append(got[0]["align_ms"])
        print("n=%d B=%d forced %2d iterations info=%d: kernel %.3f ms | last iteration kcycles: loop %.1f reduce %.1f solve %.1f; set-up %.1f" % (
            n, B, it, info, min(ms), got[0]["nn_ms"], got[0]["stats_ms"], got[0]["solve_ms"], got[0]["coarse_ms"]), flush=True)
