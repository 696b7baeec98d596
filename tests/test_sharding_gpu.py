"""GPU tests of the sharded registration path on ONE device: two contexts play two
ranks (each indexing its target slab + halo and handling its slab of the source), their
statistics blocks are summed on the device in place of the RCCL all-reduce, and the
result must equal the unsharded registration."""
import numpy as np
import pytest
import torch

from helpers import pose_error
from libwave_amd import sharding, synth

pytestmark = pytest.mark.gpu


class TwoRanksOneGpu:
    """Drives `world` GpuShardEngines on one device; sum of their stats = the all-reduce."""

    def __init__(self, ref, tgt, world, max_corr, source_pad=None):
        self.engs = [sharding.GpuShardEngine(0, ref, tgt, r, world, max_corr, source_pad=source_pad)
                     for r in range(world)]

    def begin(self, p):
        for e in self.engs:
            e.begin(p)

    def local_stats(self):
        ts = [e.local_stats() for e in self.engs]
        total = torch.stack(ts).sum(0)
        if not hasattr(self, "first"):
            self.first = [t.clone() for t in ts]   # rank-local blocks of iteration 1
        return total

    def apply(self, t):
        for e in self.engs:
            e.stats.copy_(t)
            e.apply(e.stats)

    def use_full_source(self):
        for e in self.engs:
            e.use_full_source()

    def poll(self):
        outs = [e.poll() for e in self.engs]
        for o in outs[1:]:
            assert np.array_equal(o["T"], outs[0]["T"]) and o["done"] == outs[0]["done"]
        return outs[0]


@pytest.mark.parametrize("world", [1, 2, 4])
def test_sharded_equals_unsharded(wm, ctx, world):
    ref, tgt, _ = synth.pair(40000, seed=42)
    ctx.set_source(ref)
    ctx.set_target(tgt)
    want = ctx.icp_align(max_corr=3.0, force_iterations=12, nn_method=wm.WM_NN_GRID)
    drv = sharding.ShardedIcp(TwoRanksOneGpu(ref, tgt, world, 3.0), None)
    got = drv.align(max_corr=3.0, force_iterations=12, nn_method=wm.WM_NN_GRID)
    assert got["done"] and got["iterations"] == 12 and got["n_corr"] == want["n_corr"]
    dt, ang = pose_error(got["T"], want["T"])
    assert dt <= 1e-9 and ang <= 1e-9, (dt, ang)   # differs only by summation order


def test_sharded_free_running_and_ownership(wm, ctx):
    ref, tgt, _ = synth.pair(30000, seed=7)
    ctx.set_source(ref)
    ctx.set_target(tgt)
    want = ctx.icp_align(max_corr=2.0, max_iter=60, carry_state=0)
    two = TwoRanksOneGpu(ref, tgt, 2, 2.0)
    got = sharding.ShardedIcp(two, None).align(max_corr=2.0, max_iter=60)
    assert (got["iterations"], got["state"]) == (want["iterations"], want["state"])
    dt, ang = pose_error(got["T"], want["T"])
    assert dt <= 1e-9 and ang <= 1e-9
    # every source point was handled by exactly one rank: the rank-local n sum to n_corr
    n_local = [float(t[0].item()) for t in two.first]
    ctx.nn_search(np.eye(4), 2.0, wm.WM_NN_GRID, want=False)
    n_full = ctx.icp_stats_for(np.eye(4))[0]
    assert all(n > 0 for n in n_local) and sum(n_local) == n_full


def test_narrow_source_band_is_detected_and_redone(wm, ctx):
    ref, tgt, _ = synth.pair(30000, seed=42)
    ctx.set_source(ref)
    ctx.set_target(tgt)
    want = ctx.icp_align(max_corr=3.0, force_iterations=10)
    two = TwoRanksOneGpu(ref, tgt, 2, 3.0, source_pad=0.01)   # far too narrow
    assert all(e.n_source_local < len(ref) for e in two.engs)
    got = sharding.ShardedIcp(two, None).align(max_corr=3.0, force_iterations=10)
    assert got.get("redone_with_full_source") and got["owned_violations"] == 0
    dt, ang = pose_error(got["T"], want["T"])
    assert dt <= 1e-9 and ang <= 1e-9
    ok = TwoRanksOneGpu(ref, tgt, 2, 3.0)                      # default band = max_corr
    got = sharding.ShardedIcp(ok, None).align(max_corr=3.0, force_iterations=10)
    assert not got.get("redone_with_full_source") and got["owned_violations"] == 0
    assert all(e.n_source_local < len(ref) for e in ok.engs)
    dt, ang = pose_error(got["T"], want["T"])
    assert dt <= 1e-9 and ang <= 1e-9


def test_sharded_driver_with_external_stream_world1(wm):
    ref, tgt, _ = synth.pair(20000, seed=3)
    eng = sharding.GpuShardEngine(0, ref, tgt, 0, 1, 3.0)
    got = sharding.ShardedIcp(eng, None).align(max_corr=3.0, force_iterations=8, profile=1)
    c = wm.Context(0)
    c.set_source(ref)
    c.set_target(tgt)
    want = c.icp_align(max_corr=3.0, force_iterations=8)
    dt, ang = pose_error(got["T"], want["T"])
    assert dt <= 1e-12 and ang <= 1e-12
    assert got["nn_launches"] == 8 and got["nn_ms"] > 0
