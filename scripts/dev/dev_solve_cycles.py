"""Developer: the shader-clock stamps of the last k_bins_solve of a 1M registration (start, bins + state in, statistics
expanded, solved + published)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import __graft_entry__ as g; g.build()
from libwave_amd import capi, synth
ref, tgt, T = synth.pair(1_000_000, seed=42)
dev = torch.device("cuda", 0)
d_ref, d_tgt = torch.from_numpy(ref).to(dev), torch.from_numpy(tgt).to(dev)
ctx = capi.Context(0)
for rep in range(3):
    ctx.set_source(d_ref); ctx.set_target(d_tgt)
    r = ctx.icp_align(max_corr=3.0, force_iterations=50)
    c = ctx.solve_cycles()
    print("iterations", r["iterations"], "cycles: collect %d, expand %d, solve+publish %d (before umeyama %d, umeyama %d, after %d), total %d" % (c[1] - c[0], c[2] - c[1], c[3] - c[2], c[4] - c[2], c[5] - c[4], c[3] - c[5], c[3] - c[0]))
