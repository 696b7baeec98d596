"""Cycle stamps of sampled workgroups of the certificate kernel on the bench pair (WM_CERT_PROF=1)."""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["WM_CERT_PROF"] = "1"
import numpy as np
import torch

from libwave_amd import capi, synth

ref, tgt, _ = synth.pair(1000000, seed=42)
d_ref, d_tgt = torch.from_numpy(ref).cuda(), torch.from_numpy(tgt).cuda()
ctx = capi.Context(0)
for k in range(3):
    if k == 2:
        ctx.cert_log(64)
    ctx.set_source(d_ref)
    ctx.set_target(d_tgt)
    r = ctx.icp_align(max_corr=3.0, force_iterations=50, nn_method=capi.WM_NN_GRID, profile=1, carry_state=0)
it = ctx.iteration_times() * 1e3
cnt = ctx.cert_log()
prof = ctx.cert_prof()
first = 50 - len(cnt)
names = ["state in", "loads in", "phase 1", "seeds", "passes", "coop", "stores", "sums", "-", "end p2", "end"]
print("cert launches", len(cnt), "; columns: cycles between stamps:", ", ".join(names))
for k in range(len(cnt)):
    t0 = min(prof[k, w, 0] for w in range(4) if prof[k, w, 11] != 0)
    print("it %2d: sampled workgroups start / end (cycles after the first sampled start):" % (first + k),
          "  ".join("%d: %.0f / %.0f" % (w * 256, prof[k, w, 0] - t0, prof[k, w, 11] - t0) for w in range(4) if prof[k, w, 11] != 0))
    for w in range(4):
        p = prof[k, w]
        if p[11] == 0:
            continue
        d = np.diff(p[:12])
        d[d < 0] = 0   # (stamps of skipped phases)
        seg = [p[1] - p[0], p[2] - p[1], p[3] - p[2], p[4] - p[3], p[5] - p[4], p[6] - p[5], p[7] - p[6], p[8] - p[7], 0,
               p[10] - max(p[8], p[3]), p[11] - p[10]]
        print("it %2d %5.1f us uns %6d | wg %d (uns %3d) total %6.0f :" % (first + k, it[first + k], cnt[k], w * 256, p[12], p[11] - p[0]),
              " ".join("%6.0f" % v for v in seg))
