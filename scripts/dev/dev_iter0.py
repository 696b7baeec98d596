"""Where the first (unseeded) search and the early seeded ones spend their work: per-query passes,
row chunks and trips (wm_debug_cost_log) against the query's true neighbour distance."""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import torch

from libwave_amd import capi, synth

N = 1_000_000
ITERS = 8
ref, tgt, T_gt = synth.pair(N, seed=42)
d_ref, d_tgt = torch.from_numpy(ref).cuda(), torch.from_numpy(tgt).cuda()
ctx = capi.Context(0)
ctx.set_source(d_ref)
ctx.set_target(d_tgt)
n = ctx.sizes()[0]
ctx.cost_log_arm(ITERS)
r = ctx.icp_align(max_corr=3.0, force_iterations=ITERS, nn_method=capi.WM_NN_GRID, profile=0, carry_state=0)
log = ctx.cost_log_fetch(ITERS, n)
h = r["grid_cell"]
print("cell %.3f m" % h)
for it in range(len(log)):
    c = log[it]
    trips = (c & 0xFFFF).astype(np.int64)
    chunks = ((c >> 16) & 0xFF).astype(np.int64)
    passes = ((c >> 24) & 0x7F).astype(np.int64)
    heavy = (c >> 31).astype(np.int64)
    print("it %d: heavy %d | passes hist %s" % (it, heavy.sum(), np.bincount(passes, minlength=8)[:10]))
    for p in range(1, 8):
        m = passes == p
        if m.sum():
            print("     passes=%d: %7d queries, chunks mean %.1f (total %.2e), trips mean %.1f (total %.2e)" % (
                p, m.sum(), chunks[m].mean(), chunks[m].sum(), trips[m].mean(), trips[m].sum()))
    w = np.pad(chunks, (0, (-n) % 64)).reshape(-1, 64)
    t = np.pad(trips, (0, (-n) % 64)).reshape(-1, 64)
    print("     per wave: max-lane chunks mean %.1f, sum trips/64 mean %.1f" % (w.max(1).mean(), t.sum(1).mean() / 64))
ctx.close()
# true neighbour distances at the start
c2 = capi.Context(0)
c2.set_source(d_ref)
c2.set_target(d_tgt)
idx, d2 = c2.nn_search(np.eye(4), max_corr=3.0)
d = np.sqrt(d2[idx >= 0]) / h
print("iteration 0 neighbour distance in cells: quantiles 10/25/50/75/90/99 %%:", np.round(np.quantile(d, [0.1, 0.25, 0.5, 0.75, 0.9, 0.99]), 2))
