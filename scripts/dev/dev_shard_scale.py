"""Developer probe: per-rank cost of the sharded path at world=W emulated on ONE GPU
(no all-reduce: the pose update uses this rank's partial statistics, so only timing is meaningful)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from libwave_amd import capi, sharding, synth
W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rank = int(sys.argv[2]) if len(sys.argv) > 2 else W // 2
ref, tgt, _ = synth.pair_tiled(1_000_000, W, seed=42)
eng = sharding.GpuShardEngine(0, ref, tgt, rank, W, 3.0)
print("world", W, "rank", rank, "slab", eng.lo, eng.hi, "local target", eng.n_target_local, "local source", eng.n_source_local)
drv = sharding.ShardedIcp(eng, None)
eng.n_source_total = 0   # disable the ownership check: there is no all-reduce in this probe
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.rebuild()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    r = drv.align(max_corr=3.0, force_iterations=50, profile=1)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print("rebuild %.2f ms align %.2f ms nn %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, r["nn_ms"]))
