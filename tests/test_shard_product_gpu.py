"""GPU tests of the PRODUCT's sharded registration (wm_icp_align_sharded / wm_multi, C++ loop with
the exchange inside the library; libwave_amd/csrc/wm_shard.hip).  One GPU is all a test box has,
so the ranks of a group share device 0 and the exchange is the library's host stand-in
(wm_comm_init_local) -- every other step (slab planning from the histogram, band compaction,
ownership by transformed x, per-iteration all-reduce -> identical solve on every rank) is the
code the multi-GPU run executes.  A one-rank RCCL group runs the real ncclAllReduce."""
import threading

import numpy as np
import pytest

from helpers import pose_error
from libwave_amd import synth

pytestmark = pytest.mark.gpu


def _unsharded(wm, ref, tgt, **kw):
    c = wm.Context(0)
    c.set_source(ref)
    c.set_target(tgt)
    out = c.icp_align(nn_method=wm.WM_NN_GRID, **kw)
    c.close()
    return out


@pytest.mark.parametrize("world", [2, 4])
def test_multi_emulated_equals_unsharded(wm, world):
    ref, tgt, _ = synth.pair(60000, seed=42)
    want = _unsharded(wm, ref, tgt, max_corr=3.0, force_iterations=12)
    m = wm.Multi([0] * world, emulate=True)
    got = m.icp_align(ref, tgt, max_corr=3.0, force_iterations=12, nn_method=wm.WM_NN_GRID)
    m.close()
    assert got["rc"] == 0 and got["iterations"] == 12 and got["n_corr"] == want["n_corr"]
    assert got["owned_violations"] == 0
    dt, ang = pose_error(got["T"], want["T"])
    assert dt <= 1e-9 and ang <= 1e-9, (dt, ang)  # differs only by summation order


def test_multi_free_running_same_stop(wm):
    """PCL's stopping rules, applied to the all-reduced block on every rank: same iteration count
    and state as the unsharded registration."""
    ref, tgt, _ = synth.pair(50000, seed=5, mode="resample")
    want = _unsharded(wm, ref, tgt, max_corr=2.0, max_iter=60, carry_state=0)
    m = wm.Multi([0, 0, 0], emulate=True)
    got = m.icp_align(ref, tgt, max_corr=2.0, max_iter=60, nn_method=wm.WM_NN_GRID)
    m.close()
    assert got["rc"] == want["rc"] == 0
    assert got["iterations"] == want["iterations"] and wm.CONV_NAMES[got["state"]] == want["state"]
    dt, ang = pose_error(got["T"], want["T"])
    assert dt <= 1e-8 and ang <= 1e-8


def test_ranks_on_threads_agree_bitwise(wm):
    """Per-rank entry point (what bench.py --gpus N drives, one process per rank there): every rank
    returns the SAME transform, bit for bit."""
    ref, tgt, _ = synth.pair(30000, seed=11)
    world = 3
    comms = wm.Comm.init_local(world, 0)
    outs = [None] * world

    def run(r):
        c = wm.Context(0)
        outs[r] = c.icp_align_sharded(comms[r], ref, tgt, max_corr=3.0, force_iterations=8,
                                      nn_method=wm.WM_NN_GRID)
        c.close()
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join() for t in th]
    for c in reversed(comms):
        c.close()
    assert all(o is not None and o["rc"] == 0 for o in outs)
    for o in outs[1:]:
        assert np.array_equal(o["T"], outs[0]["T"])


def test_one_rank_rccl_group_runs_the_allreduce(wm, monkeypatch):
    """ncclCommInitRank + ncclAllReduce on the context's stream (a world of one on this box)."""
    monkeypatch.setenv("WM_SHARD_FORCE", "1")
    ref, tgt, _ = synth.pair(30000, seed=3)
    want = _unsharded(wm, ref, tgt, max_corr=3.0, force_iterations=6)
    comm = wm.Comm.init_rank(0, wm.Comm.unique_id(), 0, 1)
    c = wm.Context(0)
    got = c.icp_align_sharded(comm, ref, tgt, max_corr=3.0, force_iterations=6, nn_method=wm.WM_NN_GRID)
    c.close()
    comm.close()
    assert got["rc"] == 0
    dt, ang = pose_error(got["T"], want["T"])
    assert dt <= 1e-9 and ang <= 1e-9


def test_icp_8m_in_8_slabs_full_registration(wm):
    """BASELINE configs[4]: ICP 8M<->8M as ONE registration in 8 slabs, all 50 iterations, equal to
    the unsharded registration of the same pair."""
    ref, tgt, T_gt = synth.pair_tiled(1_000_000, 8, seed=42)
    want = _unsharded(wm, ref, tgt, max_corr=3.0, force_iterations=50)
    m = wm.Multi([0] * 8, emulate=True)
    got = m.icp_align(ref, tgt, max_corr=3.0, force_iterations=50, nn_method=wm.WM_NN_GRID)
    m.close()
    assert got["rc"] == 0 and got["iterations"] == 50 and got["owned_violations"] == 0
    assert got["n_corr"] == want["n_corr"]
    dt, ang = pose_error(got["T"], want["T"])
    assert dt <= 1e-8 and ang <= 1e-9, (dt, ang)
    dt, ang = pose_error(got["T"], T_gt)
    assert dt <= 1e-3 and ang <= 1e-4, (dt, ang)


def test_ndt_device_allreduce_matches_unsharded(wm):
    """wm_ndt_set_comm: the derivative passes' sums all-reduced in HBM (stand-in exchange here)."""
    ref, tgt, _ = synth.pair(40000, seed=9, mode="resample")
    c = wm.Context(0)
    c.set_source(ref)
    c.set_target(tgt)
    want = c.ndt_align(res=1.0, step_size=0.1, max_iter=20)
    c.close()
    world = 2
    comms = wm.Comm.init_local(world, 0)
    outs = [None] * world

    def run(r):
        cc = wm.Context(0)
        cc.set_source(ref)
        cc.set_target(tgt)
        cc.ndt_set_comm(comms[r])
        outs[r] = cc.ndt_align(res=1.0, step_size=0.1, max_iter=20)
        cc.close()
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join() for t in th]
    for cm in reversed(comms):
        cm.close()
    assert np.array_equal(outs[0]["T"], outs[1]["T"])
    dt, ang = pose_error(outs[0]["T"], want["T"])
    assert dt <= 1e-6 and ang <= 1e-6, (dt, ang)
