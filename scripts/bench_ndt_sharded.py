#!/usr/bin/env python3
"""Sharded NDT (wm_ndt_set_shard) at BASELINE configs[3] sizes, run ON A GPU BOX:

  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port 29611 scripts/bench_ndt_sharded.py          # N ranks, RCCL all-reduce
  python scripts/bench_ndt_sharded.py --emulate 8                   # ONE GPU: what one rank of 8
                                                                    # costs (its share of the
                                                                    # source, sums x8 in place of
                                                                    # the all-reduce)
Every rank sets the same 2M-point clouds, evaluates its share of the source (4096-point chunks
dealt out round-robin, i.e. the whole scene at 1/N density) in each derivative pass and all-reduces the 28 pass totals.  Rank 0 prints one JSON line (median of the timed
registrations, max over ranks)."""
import json
import os

# One process per GPU: keep numpy / torch CPU thread pools small.  Their default is one thread per
# logical CPU (256 here); the pools' spinning workers burn the container's CPU quota during set-up
# and the whole process is then throttled for tens of milliseconds somewhere in the timed region
# (cgroup cpu.stat: nr_throttled) -- seen as one 50-90 ms registration per run.
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "4")
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import __graft_entry__ as g
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    emulate = int(sys.argv[sys.argv.index("--emulate") + 1]) if "--emulate" in sys.argv else 0
    n = int(sys.argv[sys.argv.index("--points") + 1]) if "--points" in sys.argv else 2_000_000
    reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 5
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if "RANK" in os.environ:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
    if rank == 0:
        g.build()
    if dist:
        dist.barrier()
    from libwave_amd import capi, sharding, synth
    ref, tgt, T_gt = synth.pair(n, seed=42)
    d_ref, d_tgt = torch.from_numpy(ref).to(dev), torch.from_numpy(tgt).to(dev)
    ctx = capi.Context(local_rank)
    if dist:
        ctx.ndt_set_shard(rank, world, sharding.make_allreduce(dist, dev))
        mode = "%d ranks, RCCL all-reduce of 28 f64 per derivative pass" % world
    elif emulate > 1:
        def times_world(vals, k, _user):   # rank 0 of `emulate`: its uniform share stands for all
            for i in range(k):
                vals[i] *= emulate
            return 0
        ctx.ndt_set_shard(0, emulate, capi.ALLREDUCE_FN(times_world))
        mode = "one rank of %d emulated on one GPU (no exchange)" % emulate
    else:
        mode = "unsharded"

    def reg():
        ctx.set_source(d_ref)
        ctx.set_target(d_tgt)
        return ctx.ndt_align(res=0.5)

    r = reg()
    torch.cuda.synchronize()
    times = []
    for _ in range(reps):
        if dist:
            dist.barrier()
        t0 = time.perf_counter()
        r = reg()
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) * 1e3)
    med = float(np.median(times))
    if dist:
        t = torch.tensor([med], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        med = float(t.item())
    if rank == 0:
        err = float(np.linalg.norm(r["T"][:3, 3] - T_gt[:3, 3])) if r["T"] is not None else None
        print(json.dumps({"config": "NDTMatcher %d<->%d, 0.5 m voxels, sharded source" % (n, n),
                          "mode": mode, "ms_per_registration": med, "ms_each_rank0": [round(t, 3) for t in times],
                          "rc": r["rc"], "iterations": r["iterations"],
                          "derivative_passes": r["evaluations"], "translation_error_m": err}))
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
