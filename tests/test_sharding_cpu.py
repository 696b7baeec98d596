"""CPU tests of the multi-GPU path's host logic: slab planning, the exactly-one-owner /
halo-exactness properties, and the sharded iteration driver over gloo with world_size 2
(the same ShardedIcp loop bench.py runs over RCCL, with an oracle-backed engine standing
in for the HIP kernels; the solve + stopping rules are the product's own host function)."""
import os
import sys

import numpy as np
import pytest

from helpers import pose_error, svd_stats_numpy
from libwave_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plan_slabs_properties():
    from libwave_amd import sharding
    tgt = synth.scene(50000, seed=3)
    for world in (1, 2, 4, 8):
        slabs = sharding.plan_slabs(tgt, world)
        assert len(slabs) == world and slabs[0][0] == -np.inf and slabs[-1][1] == np.inf
        for (l0, h0), (l1, h1) in zip(slabs[:-1], slabs[1:]):
            assert h0 == l1 and l0 < h0
            assert np.float32(h0) == h0          # representable in the device's compare type
        x = tgt[:, 0]
        owner = np.zeros(len(x), int)
        for lo, hi in slabs:
            owner += ((x >= lo) & (x < hi)).astype(int)
        assert (owner == 1).all()                # every point has exactly one owner
        counts = [int(((x >= lo) & (x < hi)).sum()) for lo, hi in slabs]
        assert max(counts) - min(counts) <= 0.02 * len(x) + 2


def test_halo_makes_sharded_nn_exact(oracle):
    from libwave_amd import sharding
    ref, tgt, _ = synth.pair(20000, seed=5)
    max_corr = 1.5
    full_idx, full_d2 = oracle.KdTree(tgt).nn(ref)
    keep = full_d2.astype(np.float64) <= max_corr ** 2
    got = np.full(len(ref), -1)
    for lo, hi in sharding.plan_slabs(tgt, 4):
        mask = sharding.slab_target_mask(tgt, lo, hi, max_corr * (1 + 1e-6) + 1e-4)
        gidx = np.nonzero(mask)[0]
        mine = (ref[:, 0] >= lo) & (ref[:, 0] < hi)
        li, ld2 = oracle.KdTree(tgt[mask]).nn(ref[mine])
        ok = ld2.astype(np.float64) <= max_corr ** 2
        got[np.nonzero(mine)[0][ok]] = gidx[li[ok]]
    assert np.array_equal(got, np.where(keep, full_idx, -1))


class OracleShardEngine:
    """Same interface as sharding.GpuShardEngine, kernels replaced by the CPU oracle."""

    def __init__(self, ref, tgt, rank, world, max_corr, oracle, capi, source_pad=None):
        import torch
        from libwave_amd import sharding
        self.torch, self.O, self.capi = torch, oracle, capi
        self.lo, self.hi = sharding.plan_slabs(tgt, world)[rank]
        self.ref_full = ref
        self.n_total = len(ref)
        pad = max_corr if source_pad is None else source_pad
        self.ref = ref if world == 1 else ref[sharding.slab_source_mask(ref, self.lo, self.hi, pad)]
        mask = sharding.slab_target_mask(tgt, self.lo, self.hi, max_corr * (1 + 1e-6) + 1e-4)
        self.tgt = tgt[mask]
        self.tree = oracle.KdTree(self.tgt)
        self.max_corr = max_corr

    def use_full_source(self):
        self.ref = self.ref_full

    def begin(self, params):
        self.host = self.capi.HostIcp(params, self.n_total)

    def local_stats(self):
        T = self.host.get()["T"]
        moved = self.O.transform_cloud_f(self.ref, T.astype(np.float32))
        mine = (moved[:, 0] >= np.float32(self.lo)) & (moved[:, 0] < np.float32(self.hi))
        st = np.zeros(32)
        if mine.any() and len(self.tgt):
            idx, d2 = self.tree.nn(moved[mine])
            ok = d2.astype(np.float64) <= self.max_corr ** 2
            st = svd_stats_numpy(moved[mine][ok], self.tgt[idx[ok]], d2[ok])
        st[31] = mine.sum()          # source points this rank handled (ownership check)
        return self.torch.from_numpy(st)

    def apply(self, t):
        self.host.apply(t.numpy())

    def poll(self):
        return self.host.get()


def _worker(rank, world, port, n, q, source_pad=None):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from libwave_amd import capi, sharding
    from oracle import oracle_py as O
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ref, tgt, _ = synth.pair(n, seed=42)
        eng = OracleShardEngine(ref, tgt, rank, world, 3.0, O, capi, source_pad)
        out = {"n_local": len(eng.ref)}
        for name, kw in (("forced", dict(force_iterations=6)), ("free", dict(max_iter=40))):
            r = sharding.ShardedIcp(eng, dist).align(max_corr=3.0, **kw)
            out[name] = (r["T"].tolist(), r["iterations"], r["state"], r["n_corr"])
            out[name + "_redone"] = bool(r.get("redone_with_full_source"))
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("source_pad,port", [(None, 29731), (0.01, 29741)])
def test_sharded_icp_over_gloo_matches_unsharded_oracle(oracle, wm, source_pad, port):
    """source_pad=None: each rank holds the source points within max_corr of its slab.
    source_pad=0.01: a band far too narrow -> the ownership count exposes the lost points and
    the driver redoes the registration with full source clouds; the result must not change."""
    import multiprocessing as mp
    n, world = 6000, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q, source_pad))
             for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref, tgt, _ = synth.pair(n, seed=42)
    assert res[0]["n_local"] < n and res[1]["n_local"] < n     # bands, not the full cloud
    if source_pad is not None:
        assert res[0]["forced_redone"] and res[1]["forced_redone"]
    else:
        assert not res[0]["forced_redone"] and not res[0]["free_redone"]
    for name, kw in (("forced", dict(force_iterations=6)), ("free", dict(max_iter=40))):
        want = oracle.icp_align(ref, tgt, max_corr=3.0, incremental_float=0, **kw)
        T0, it0, st0, nc0 = res[0][name]
        T1, it1, st1, nc1 = res[1][name]
        assert np.array_equal(np.array(T0), np.array(T1))      # every rank solves identically
        assert (it0, st0, nc0) == (it1, st1, nc1) == (want["iterations"], want["state"],
                                                        want["n_corr"])
        dt, ang = pose_error(np.array(T0), want["T"])
        assert dt <= 1e-9 and ang <= 1e-9, (dt, ang)


def test_host_icp_state_machine_matches_oracle_single_rank(oracle, wm):
    from libwave_amd import sharding
    ref, tgt, _ = synth.pair(4000, seed=9)
    eng = OracleShardEngine(ref, tgt, 0, 1, 3.0, oracle, wm)
    got = sharding.ShardedIcp(eng, None).align(max_corr=3.0, max_iter=50)
    want = oracle.icp_align(ref, tgt, max_corr=3.0, max_iter=50, incremental_float=0)
    assert (got["iterations"], got["state"]) == (want["iterations"], want["state"])
    dt, ang = pose_error(got["T"], want["T"])
    assert dt <= 1e-9 and ang <= 1e-9


def _reduce_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import ctypes as C
    import torch.distributed as dist
    from libwave_amd import sharding
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cb = sharding.make_allreduce(dist)          # the callback wm_ndt_set_shard is given
        out = []
        for call in range(3):                        # as many calls as derivative passes
            vals = (C.c_double * 28)(*[(rank + 1) * 0.1 * (k + 1) + call for k in range(28)])
            rc = cb(vals, 28, None)
            out.append((rc, list(vals)))
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_ndt_allreduce_callback_over_gloo_world2():
    """The reduce callback of the sharded NDT (wm_ndt_set_shard): called through its C signature
    on two gloo ranks, it must leave the same sums -- bit for bit -- on both."""
    import multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_reduce_worker, args=(r, world, 29761, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=100) for _ in range(world))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    for call in range(3):
        rc0, v0 = res[0][call]
        rc1, v1 = res[1][call]
        assert rc0 == 0 and rc1 == 0 and v0 == v1
        want = [(0.1 * (k + 1) + call) + (0.2 * (k + 1) + call) for k in range(28)]
        assert np.allclose(v0, want, rtol=1e-15, atol=0)


def test_thread_group_reduce_sums_in_rank_order():
    import ctypes as C
    import threading
    from libwave_amd import sharding
    g = sharding.ThreadGroupReduce(3)
    out = [None] * 3

    def run(r):
        cb = g.callback(r)
        for call in range(4):
            a = (C.c_double * 5)(*[10.0 ** r + k + call for k in range(5)])
            assert cb(a, 5, None) == 0
            out[r] = list(a)
    ts = [threading.Thread(target=run, args=(r,)) for r in range(3)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(30)
    assert out[0] == out[1] == out[2] == [111.0 + 3 * (k + 3) for k in range(5)]
