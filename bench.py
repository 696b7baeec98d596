#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline metric on MI355X.

  metric : point-cloud registrations/sec (1M<->1M pts, ICP, 50 iterations)
  step   : ONE full registration through the C ABI with both clouds already resident
           in HBM: wm_set_source (Morton order) + wm_set_target (grid build) +
           wm_icp_align (50 forced iterations: correspondence search, statistics
           reduction, Umeyama solve, all on device).
  N > 1  : one process per GPU (torch.distributed / RCCL).  The registration is sharded:
           every rank indexes one x-slab of the target (+ max_corr halo), handles the
           source points that fall into its slab, and the 32-double statistics block is
           all-reduced once per iteration (weak scaling: N x 1M points per cloud).

Prints ONE JSON line on rank 0 (contract in the task statement), with two extra
objects: "roofline" (correspondence kernel vs the HBM roof) and "cpu_baseline"
(the CPU oracle timed on this box's host cores, N == 1 only).
"""
import argparse
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

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s copy-measured)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--points", type=int, default=1_000_000, help="points per cloud PER GPU")
    ap.add_argument("--iters", type=int, default=50, help="forced ICP iterations per registration")
    ap.add_argument("--max-corr", type=float, default=3.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=20, help="oracle iterations timed for the baseline")
    return ap.parse_args()


def cpu_baseline(ref, tgt, iters_full, max_corr, cpu_iters):
    """The CPU oracle (oracle/: kd-tree exact NN + Umeyama, single thread like PCL) on a
    bounded sample of the same workload: tree build + `cpu_iters` iterations are timed,
    then scaled to the `iters_full`-iteration registration."""
    from oracle import oracle_py as O
    O.lib()
    t0 = time.perf_counter()
    O.KdTree(tgt)
    t_build = time.perf_counter() - t0
    t0 = time.perf_counter()
    O.icp_align(ref, tgt, max_corr=max_corr, force_iterations=cpu_iters)
    t_run = time.perf_counter() - t0
    per_iter = max(t_run - t_build, 1e-9) / cpu_iters
    t_full = t_build + per_iter * iters_full
    return {
        "value": 1.0 / t_full, "unit": "registrations/s", "cores": 1, "kind": "port",
        "sample": "same %d<->%d clouds; kd-tree build (%.2f s) + %d of %d iterations timed "
                  "(%.3f s/iter), scaled to %d iterations" % (len(ref), len(tgt), t_build,
                                                              cpu_iters, iters_full, per_iter,
                                                              iters_full),
        "seconds_per_registration": t_full, "host_cpus": os.cpu_count(),
    }


def measured_copy_bandwidth(torch, dev):
    """Device-to-device copy rate of this very box (GB/s, read + write bytes): the practical HBM
    ceiling to hold next to the 8 TB/s specification (SURVEY 8(d) asks for it in the same run)."""
    n = 1 << 28  # 1 GiB of f32 in, 1 GiB out
    a = torch.empty(n, dtype=torch.float32, device=dev).normal_()
    b = torch.empty_like(a)
    for _ in range(2):
        b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    reps = 10
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    del a, b
    return 2.0 * n * 4 / (ms * 1e-3) / 1e9


def pmc_traffic_bytes():
    """HBM bytes per launch of the correspondence kernel from the committed PMC summary
    (profiles/pmc_latest.json, written by scripts/gpu_pmc.sh + scripts/pmc_to_json.py from
    separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same bench command).
    FETCH_SIZE is doubled (MI355X_MICROARCH.md: gfx950 tallies 128-B requests at 64 B for wide
    coalesced 16-B/lane reads); WRITE_SIZE is taken as reported.  None if not available."""
    path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    try:
        with open(path) as f:
            d = json.load(f)["k_nn_grid"]
        return (2.0 * d["FETCH_SIZE_kb_per_dispatch"] + d["WRITE_SIZE_kb_per_dispatch"]) * 1024.0
    except Exception:
        return None


def main():
    a = parse()
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with %d ranks" % (a.gpus, a.gpus))
        a.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback exists)")
    torch.cuda.set_device(local_rank)
    dist = None
    force_sharded = os.environ.get("WM_BENCH_FORCE_SHARDED") == "1"  # plumbing check at N=1
    if world > 1 or (force_sharded and "RANK" in os.environ):
        # (a one-rank group under torch.distributed.run still sends every iteration's block
        # through RCCL: the whole N > 1 code path on a single GPU)
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import __graft_entry__ as g
    if rank == 0:
        g.build()
    if dist:
        dist.barrier()
    from libwave_amd import capi, synth

    n_total = a.points * world
    # weak scaling: `world` tiles of the base scene side by side (same density, longer map);
    # world == 1 is the plain 1M<->1M pair of BASELINE configs[1]
    ref, tgt, T_gt = synth.pair_tiled(a.points, world, seed=42)
    dev = torch.device("cuda", local_rank)

    if world == 1 and not force_sharded:
        d_ref = torch.from_numpy(ref).to(dev)
        d_tgt = torch.from_numpy(tgt).to(dev)
        ctx = capi.Context(local_rank)
        torch.cuda.synchronize()

        def step(profile):
            ctx.set_source(d_ref)
            ctx.set_target(d_tgt)
            return ctx.icp_align(max_corr=a.max_corr, force_iterations=a.iters,
                                 nn_method=capi.WM_NN_GRID, profile=profile, carry_state=0)
        parallelism = "single"
    else:
        from libwave_amd import sharding
        eng = sharding.GpuShardEngine(local_rank, ref, tgt, rank, world, a.max_corr)
        drv = sharding.ShardedIcp(eng, dist, dev)

        def step(profile):
            eng.rebuild()
            return drv.align(max_corr=a.max_corr, force_iterations=a.iters, profile=profile)
        parallelism = "target-slab x%d + all-reduce(32 f64)/iter" % world

    for _ in range(a.warmup):
        r = step(1)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    t0 = time.perf_counter()
    nn_ms = 0.0
    nn_launches = 0
    step_ms = []
    for k in range(a.steps):
        # HIP events around every launch of the correspondence kernel cost a barrier packet
        # each, so they bracket the launches of ONE timed step in three (at least the last);
        # the others run exactly as a caller's registration would
        timed = (k % 3 == 2) or k == a.steps - 1
        t_step = time.perf_counter()
        r = step(1 if timed else 0)
        step_ms.append((time.perf_counter() - t_step) * 1e3)  # (align blocks: host time = step time)
        if timed:
            nn_ms += r["nn_ms"]
            nn_launches += r["nn_launches"]
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        T = r["T"]
        err_t = float(np.linalg.norm(T[:3, 3] - T_gt[:3, 3]))
        # whole-job throughput in 1M-point registration equivalents (weak scaling: the
        # N-GPU job registers an N x 1M-point pair, i.e. N units of the metric's size)
        regs_per_s_raw = a.steps / elapsed
        value = regs_per_s_raw * (n_total / 1_000_000.0)
        nn_us = nn_ms / max(nn_launches, 1) * 1e3
        pts_per_launch = a.points  # queries handled by one rank's launch
        alg_bytes = 32.0 * pts_per_launch  # 12N + 12M + 8N with N = M
        achieved = alg_bytes / (nn_us * 1e-6) / 1e9 if nn_us > 0 else 0.0
        out = {
            "metric": "point-cloud registrations/sec (1M<->1M pts, ICP)",
            "value": value, "unit": "registrations/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": "ICPMatcher %d<->%d synthetic XYZ clouds (BASELINE configs[%d]), "
                            "%d forced iterations, max_corr=%g, res=-1; step = set_source + "
                            "set_target (index build) + align" % (n_total, n_total,
                                                                 1 if world == 1 else 4,
                                                                 a.iters, a.max_corr),
                "arithmetic": "f32 points and distances, f64 reductions and solve",
                "scene": "base scene x%d tiled along x at constant density; T_gt rotation / %d" % (world, world),
                "points_per_cloud_total": n_total, "points_per_gpu": a.points,
                "iterations": a.iters, "parallelism": parallelism,
                "registrations_per_s_raw": regs_per_s_raw,
                "icp_iterations_per_s": regs_per_s_raw * a.iters,
                "final_translation_error_m": err_t, "grid_cell_m": r.get("grid_cell"),
                "deferred_queries_per_registration": r.get("deferred"),
                "ms_each_step_rank0": [round(t, 3) for t in step_ms],
            },
            "roofline": {
                "bound": "hbm", "kernel": "wm::k_nn_grid (correspondence search)",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": pmc_traffic_bytes() if world == 1 else None,
                "peak_measured_copy": measured_copy_bandwidth(torch, dev) if world == 1 else None,
                "algorithmic_bytes_per_launch": alg_bytes,
                "avg_launch_us": nn_us, "launches_timed": nn_launches,
                "note": "the kernel streams 64 B/point (source, previous key + match in; key + match "
                        "out) and gathers its candidates out of L1/L2 (hit rate ~89 %); per-iteration "
                        "PMC (profiles/r01d_pmc_k_nn_grid_per_iteration.csv): VALU ~50 % and texture "
                        "addresser ~59 % busy, ~66 % of lanes active -- a dependent chain of small "
                        "gathers, not an HBM stream",
            },
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(ref, tgt, a.iters, a.max_corr, a.cpu_iters)
            out["config"]["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
