"""Developer: the 64-ring 1M pair (50 forced iterations) under different level-0 cell sizes (the automatic choice first)."""
import os, sys, time
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "4")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import __graft_entry__ as g; g.build()
from libwave_amd import capi, synth
dev = torch.device("cuda", 0)
pattern = sys.argv[1] if len(sys.argv) > 1 else "rings"
ref, tgt, T = synth.pair(1_000_000, seed=42, pattern=pattern)
d_ref, d_tgt = torch.from_numpy(ref).to(dev), torch.from_numpy(tgt).to(dev)
def run(ctx):
    ts = []
    for rep in range(8):
        t0 = time.perf_counter()
        ctx.set_source(d_ref); ctx.set_target(d_tgt)
        r = ctx.icp_align(max_corr=3.0, force_iterations=50)
        ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts[3:])), r
ctx = capi.Context(0)
ms, r = run(ctx)
h0 = r["grid_cell"]
print("%s: automatic cell %.4f m: %.3f ms" % (pattern, h0, ms), flush=True)
for f in (0.4, 0.5, 0.6, 0.7, 0.85, 1.2, 1.5):
    c = capi.Context(0)
    c.set_grid_cell(h0 * f)
    ms, r2 = run(c)
    same = np.array_equal(r2["T"], r["T"])
    print("   cell x %.2f = %.4f m: %.3f ms   same transform %s" % (f, r2["grid_cell"], ms, same), flush=True)
