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


def test_multi_redoes_with_full_source_when_points_leave_their_bands(wm):
    """A source 2 m off along the slab axis: the registration pulls points out of the bands their ranks were given
    (max_corr wide), the all-reduced count of handled points says so, and every rank redoes the registration with the
    whole source (wm_icp_stats.shard_attempts = 2: the second selection of k_band_count / k_band_write) -- to the
    unsharded result."""
    ref, tgt, _ = synth.pair(60000, seed=21)
    ref = ref.copy()
    ref[:, 0] -= 2.0
    want = _unsharded(wm, ref, tgt, max_corr=3.0, force_iterations=30)
    m = wm.Multi([0] * 4, emulate=True)
    got = m.icp_align(ref, tgt, max_corr=3.0, force_iterations=30, nn_method=wm.WM_NN_GRID)
    m.close()
    assert got["rc"] == want["rc"] == 0 and got["shard_attempts"] == 2 and got["owned_violations"] == 0
    assert got["n_corr"] == want["n_corr"]
    dt, ang = pose_error(got["T"], want["T"])
    assert dt <= 1e-9 and ang <= 1e-9, (dt, ang)
    assert abs(want["T"][0, 3]) > 1.5  # (the cloud did move by most of the offset)


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
    # the loop's exchange ran through the rank's mailbox, inside the solve kernel (wm_xchg.hpp)
    assert got["exchange_in_kernel"] == 1 and got["rccl_ranks"] == 1


def test_one_rank_rccl_group_collective_exchange_gives_the_same_bits(wm, monkeypatch):
    """WM_COMM_P2P=0: ncclAllReduce between a rank's sums and its solve (three launches per iteration's tail)
    and the in-kernel exchange (one) are the same arithmetic: identical transforms."""
    monkeypatch.setenv("WM_SHARD_FORCE", "1")
    ref, tgt, _ = synth.pair(30000, seed=3)
    outs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("WM_COMM_P2P", mode)
        comm = wm.Comm.init_rank(0, wm.Comm.unique_id(), 0, 1)
        c = wm.Context(0)
        outs[mode] = c.icp_align_sharded(comm, ref, tgt, max_corr=3.0, max_iter=40, nn_method=wm.WM_NN_GRID)
        c.close()
        comm.close()
    assert outs["1"]["rc"] == outs["0"]["rc"] == 0
    assert outs["1"]["exchange_in_kernel"] == 1 and outs["0"]["exchange_in_kernel"] == 0
    assert outs["1"]["iterations"] == outs["0"]["iterations"]
    assert np.array_equal(outs["1"]["T"], outs["0"]["T"])


_MAILBOX_PAIR = r"""
import os, sys, threading
import numpy as np
sys.path.insert(0, sys.argv[1])
from libwave_amd import capi as wm, synth
ref, tgt, _ = synth.pair(40000, seed=17)
world = 2
# the SAME two contexts for both runs: the first leaves every buffer allocated (on one GPU a rank's first-call
# hipMalloc / hipFree waits for the whole device -- for the other rank's solve kernel, which polls for this rank's block)
ctxs = [wm.Context(0) for _ in range(world)]
def run_group():
    comms = wm.Comm.init_local(world, 0)
    outs = [None] * world
    def run(r):
        outs[r] = ctxs[r].icp_align_sharded(comms[r], ref, tgt, max_corr=3.0, force_iterations=15, nn_method=wm.WM_NN_GRID)
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join() for t in th]
    for c in reversed(comms):
        c.close()
    return outs
want = run_group()
os.environ["WM_COMM_P2P_LOCAL"] = "1"
os.environ["WM_COMM_P2P_TIMEOUT_MS"] = "3000"
got = run_group()
assert all(o is not None and o["rc"] == 0 for o in want + got)
assert all(o["exchange_in_kernel"] == 0 for o in want) and all(o["exchange_in_kernel"] == 1 for o in got)
assert np.array_equal(got[0]["T"], got[1]["T"])
assert np.array_equal(got[0]["T"], want[0]["T"])
assert got[0]["owned_violations"] == 0 and got[0]["n_corr"] == want[0]["n_corr"]
print("MAILBOX PAIR OK")
"""


def test_two_ranks_exchange_through_mailboxes():
    """The mailbox protocol between two ranks that really run side by side (two threads, two streams, one GPU:
    WM_COMM_P2P_LOCAL=1): each solve kernel writes its block into both mailboxes and polls its own for both --
    the same transform, bit for bit, on both ranks and as the host-side stand-in's; a time limit turns a stall
    into an error instead of a hang.  In a process of its own with a hardware queue per stream: two ranks on ONE
    GPU whose streams share a hardware queue would queue one rank's kernels behind the other's polling solve kernel
    (ranks on GPUs of their own cannot meet that)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GPU_MAX_HW_QUEUES="8")
    env.pop("WM_COMM_P2P_LOCAL", None)
    out = subprocess.run([sys.executable, "-c", _MAILBOX_PAIR, root], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "MAILBOX PAIR OK" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])


_MAILBOX_FALLBACK = r"""
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
os.environ["WM_COMM_P2P_LOCAL"] = "1"
os.environ["WM_COMM_P2P_TIMEOUT_MS"] = "0"   # a peer's block that is not there at once counts as never coming
from libwave_amd import capi as wm, synth
ref, tgt, _ = synth.pair(40000, seed=17)
c = wm.Context(0); c.set_source(ref); c.set_target(tgt)
want = c.icp_align(max_corr=3.0, force_iterations=15, nn_method=wm.WM_NN_GRID); c.close()
m = wm.Multi([0, 0, 0], emulate=True)   # fresh contexts: on ONE GPU their first-call allocations stall the mailbox exchange
got = m.icp_align(ref, tgt, max_corr=3.0, force_iterations=15, nn_method=wm.WM_NN_GRID)
again = m.icp_align(ref, tgt, max_corr=3.0, force_iterations=15, nn_method=wm.WM_NN_GRID)
m.close()
assert got["rc"] == 0 and again["rc"] == 0, (got["rc"], again["rc"])
assert np.abs(got["T"] - want["T"]).max() <= 1e-9 and np.array_equal(got["T"], again["T"])
assert got["exchange_in_kernel"] == 0 and again["exchange_in_kernel"] == 0   # (the mailboxes were dropped for good)
print("FALLBACK OK")
"""


def test_multi_goes_back_to_the_collective_exchange_when_the_mailboxes_fail():
    """wm_multi_icp_match: a mailbox exchange that fails (here: three ranks on one GPU and a time limit of zero -- a
    peer's block that is not in the mailbox at once counts as never coming) does not fail the registration: every rank drops its mailboxes, the communicators stay, and the
    registration is run once more with the group's other exchange -- the unsharded result either way."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GPU_MAX_HW_QUEUES="8")
    out = subprocess.run([sys.executable, "-c", _MAILBOX_FALLBACK, root], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "FALLBACK OK" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])


_MAILBOX_DISAGREE = r"""
import os, sys, threading
import numpy as np
sys.path.insert(0, sys.argv[1])
os.environ["WM_COMM_P2P_LOCAL"] = "1"
os.environ["WM_COMM_P2P_TIMEOUT_MS"] = "4000"
from libwave_amd import capi as wm, synth
ref, tgt, _ = synth.pair(40000, seed=17)
world = 2
ctxs = [wm.Context(0) for _ in range(world)]
comms = wm.Comm.init_local(world, 0)
assert all(c.mailboxes for c in comms)
def run_group(iters):
    outs = [None] * world
    def run(r):
        try:
            outs[r] = ctxs[r].icp_align_sharded(comms[r], ref, tgt, max_corr=3.0, force_iterations=iters, nn_method=wm.WM_NN_GRID)
        except wm.WmError as e:
            outs[r] = e
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join() for t in th]
    return outs
warm = run_group(3)          # (buffers grown: no first-call allocation stalls the exchange below)
assert all(isinstance(o, dict) and o["rc"] == 0 and o["exchange_in_kernel"] == 1 for o in warm), warm
# ONE rank gives up at once (a time limit of zero), the other would wait 4 s: without the commit round rank 1 could
# end its registration with an error and rank 0 well -- and the next registration would find them on different exchanges
comms[1].set_exchange_timeout_ms(0)
bad = run_group(40)          # (40 rounds: rank 1 is the first to reach one of them, finds nothing, gives up)
assert all(isinstance(o, wm.WmError) for o in bad), bad        # BOTH ranks fail this registration ...
assert not any(c.mailboxes for c in comms)                      # ... and BOTH have left the mailboxes
good = run_group(8)                                             # the group goes on with the collective exchange
assert all(isinstance(o, dict) and o["rc"] == 0 and o["exchange_in_kernel"] == 0 for o in good), good
assert np.array_equal(good[0]["T"], good[1]["T"])
for c in reversed(comms):
    c.close()
print("DISAGREE OK")
"""


def test_ranks_agree_on_a_failed_mailbox_exchange():
    """a mailbox time-out used to be every rank's private affair -- one rank could fail a
    registration and fall back to ncclAllReduce while its peer, which had still received every block, ended well and
    kept polling mailboxes: the next registration hung.  The commit round at the end of the loop (k_xchg_commit) makes
    the verdict the group's: here one of two ranks gets a time limit of zero -- both fail that registration, both leave
    the mailboxes, and the next registration runs on the collective exchange on both."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GPU_MAX_HW_QUEUES="8")
    out = subprocess.run([sys.executable, "-c", _MAILBOX_DISAGREE, root], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "DISAGREE OK" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])


def test_icp_8m_in_8_slabs_full_registration(wm):
    """BASELINE configs[4]: ICP 8M<->8M as ONE registration in 8 slabs, all 50 iterations, equal to
    the unsharded registration of the same pair."""
    ref, tgt, T_gt = synth.pair_tiled(1_000_000, 8, seed=42)
    want = _unsharded(wm, ref, tgt, max_corr=3.0, force_iterations=50)
    m = wm.Multi([0] * 8, emulate=True)
    first = m.icp_align(ref, tgt, max_corr=3.0, force_iterations=50, nn_method=wm.WM_NN_GRID)
    # (the budget below is read off a SECOND call: the first one also pays for growing every rank's buffers)
    got = m.icp_align(ref, tgt, max_corr=3.0, force_iterations=50, nn_method=wm.WM_NN_GRID)
    m.close()
    assert first["rc"] == 0 and np.array_equal(first["T"], got["T"])
    assert got["rc"] == 0 and got["iterations"] == 50 and got["owned_violations"] == 0
    assert got["n_corr"] == want["n_corr"]
    dt, ang = pose_error(got["T"], want["T"])
    assert dt <= 1e-8 and ang <= 1e-9, (dt, ang)
    dt, ang = pose_error(got["T"], T_gt)
    assert dt <= 1e-3 and ang <= 1e-4, (dt, ang)
    # rank 0's budget: what does NOT shrink with the number of ranks (planning over the sub-sample,
    # selecting the rank's bands out of the full clouds) against what does (its local index, its
    # iterations); eight ranks share ONE GPU here, so every figure is inflated alike
    fixed = got["plan_ms"] + got["compact_ms"]
    total = fixed + got["index_ms"] + got["iter_ms"]
    print("8M in 8 slabs, rank 0: plan %.2f + select %.2f ms (do not scale) | index %.2f + iterations %.2f ms | "
          "local clouds %d / %d points | non-scaling share %.1f %%" % (
              got["plan_ms"], got["compact_ms"], got["index_ms"], got["iter_ms"], got["n_tgt_local"],
              got["n_src_local"], 100.0 * fixed / total))
    assert got["shard_attempts"] == 1 and 0.9e6 < got["n_tgt_local"] < 1.3e6 and 0.9e6 < got["n_src_local"] < 1.3e6
    # (no bar on the share itself: with eight ranks time-sharing one GPU, rank 0's planning time includes
    # however long it waits for the slowest rank at the planning all-reduce -- 4 ms or 28 ms from run to run)


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


def test_sharded_certified_iterations_equal_unsharded(wm):
    """Long enough for the certificate kernel to take over on every rank (ownership by transformed x
    inside it): the sharded registration still equals the unsharded one, ranks agree bit for bit, and
    the per-phase budget is filled in."""
    ref, tgt, _ = synth.pair(80000, seed=23, mode="resample")
    want = _unsharded(wm, ref, tgt, max_corr=3.0, force_iterations=40, carry_state=0)
    assert want["cert_launches"] > 0
    world = 3
    comms = wm.Comm.init_local(world, 0)
    outs = [None] * world

    def run(r):
        c = wm.Context(0)
        outs[r] = c.icp_align_sharded(comms[r], ref, tgt, max_corr=3.0, force_iterations=40,
                                      nn_method=wm.WM_NN_GRID, carry_state=0, profile=1)
        c.close()
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join() for t in th]
    for c in reversed(comms):
        c.close()
    for o in outs:
        assert o["rc"] == 0 and o["owned_violations"] == 0 and o["shard_attempts"] == 1
        assert np.array_equal(o["T"], outs[0]["T"])
        assert o["n_corr"] == want["n_corr"]
        assert o["plan_ms"] > 0 and o["compact_ms"] > 0 and o["index_ms"] > 0 and o["iter_ms"] > 0
        assert 0 < o["n_tgt_local"] < len(tgt) and 0 < o["n_src_local"] < len(ref)
    assert any(o["cert_launches"] > 0 for o in outs)
    dt, ang = pose_error(outs[0]["T"], want["T"])
    assert dt <= 1e-9 and ang <= 1e-9, (dt, ang)


def test_sharded_clouds_with_nonfinite_points_and_strides(wm):
    """Raw-cloud planning: xyz and xyzw strides, NaN / inf points dropped, the cloud's finite count
    agreed on through the all-reduced block (no ownership violation, one attempt)."""
    ref, tgt, _ = synth.pair(40000, seed=31, mode="resample")
    ref = ref.copy()
    tgt = tgt.copy()
    ref[::101, 0] = np.nan
    ref[5::103, 2] = np.inf
    tgt[::97, 1] = np.nan
    want = _unsharded(wm, ref, tgt, max_corr=3.0, force_iterations=10, carry_state=0)
    for cols in (3, 4):
        r4 = np.zeros((len(ref), cols), np.float32)
        t4 = np.zeros((len(tgt), cols), np.float32)
        r4[:, :3] = ref
        t4[:, :3] = tgt
        m = wm.Multi([0, 0], emulate=True)
        got = m.icp_align(r4, t4, max_corr=3.0, force_iterations=10, nn_method=wm.WM_NN_GRID)
        m.close()
        assert got["rc"] == 0 and got["owned_violations"] == 0
        assert got["n_corr"] == want["n_corr"]
        dt, ang = pose_error(got["T"], want["T"])
        assert dt <= 1e-9 and ang <= 1e-9


def test_bench_sharded_under_torch_distributed_run():
    """The driver's multi-GPU launch line, on the one GPU a test box has: a one-rank RCCL group under
    torch.distributed.run drives ncclCommInitRank / the per-iteration ncclAllReduce / the whole sharded
    path (WM_BENCH_FORCE_SHARDED=1), and the line explains itself (per-phase budget, RCCL rank count)."""
    import json
    import os
    import subprocess
    import sys
    import socket
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WM_BENCH_FORCE_SHARDED="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    with socket.socket() as s:  # a port nobody holds right now
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"),
           "--gpus", "1", "--steps", "3", "--warmup", "1", "--points", "200000", "--no-cpu-baseline",
           "--no-other-configs"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    sh = d["config"]["sharding"]
    assert d["n_gpus"] == 1 and d["value"] > 0, line
    assert sh["rccl_ranks"] == 1 and sh["owned_violations"] == 0 and sh["shard_attempts"] == 1, sh
    # the exchange ran inside the solve kernel (the rank's mailbox): no collective launches to time
    assert sh["allreduce_us_isolated"] > 0 and sh["exchange_in_kernel"] == 1 and sh["allreduce_ms"] == 0, sh
    assert sh["plan_ms"] > 0 and sh["compact_ms"] > 0 and sh["index_ms"] > 0 and sh["iter_ms"] > 0, sh
    assert d["config"]["final_translation_error_m"] < 2e-3, line
    # ... and the line names the exchange that ran and carries a second, short timing of the SAME registration with
    # north_star's exchange, ncclAllReduce: the first real multi-GPU record compares the two
    assert sh["exchange_that_ran"] == "mailboxes" and sh["exchange_fell_back_to_collective"] is False, sh
    ce = sh["collective_exchange"]
    assert ce.get("ms_per_step", 0) > 0 and ce["exchange_in_kernel"] == 0 and ce["same_transform_as_the_mailbox_run"] is True, ce
    assert d["config"]["ms_per_step_rccl_allreduce_exchange"] > 0 and d["config"]["ms_per_step_mailbox_exchange"] > 0


@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("res,steps", [(-1.0, 0), (0.1, 0), (0.1, 3)])
def test_multi_match_and_estimators_equal_one_device(wm, world, res, steps):
    """ICPMatcher::match() over a group of devices -- full resolution, voxel-filtered, and the reference's
    DEFAULT parameters (res 0.1, three coarser scales: icp.hpp:54,59, icp.cpp:77-104) -- and
    estimateInfo() after it (icp.cpp:135-142 -> impl/multi_matcher_impl.hpp:48): same transform, same
    stop, same three information matrices as one device."""
    ref, tgt, _ = synth.pair(60000, seed=77, mode="resample")
    kw = dict(max_corr=3.0, max_iter=100, nn_method=wm.WM_NN_GRID)
    c = wm.Context(0)
    one = c.icp_match(ref, tgt, res=res, multiscale_steps=steps, carry_state=0, **kw)
    infos = {}
    for name, method in (("lum", wm.WM_INFO_LUM), ("censi", wm.WM_INFO_CENSI), ("lumold", wm.WM_INFO_LUMOLD)):
        infos[name] = c.icp_info(method, one["T"], max_corr=3.0)
    c.close()
    assert one["rc"] == 0
    m = wm.Multi([0] * world, emulate=True)
    got = m.icp_match(ref, tgt, res=res, multiscale_steps=steps, carry_state=0, **kw)
    assert got["rc"] == 0 and got["owned_violations"] == 0
    assert got["iterations"] == one["iterations"] and got["n_corr"] == one["n_corr"]
    dt, ang = pose_error(got["T"], one["T"])
    assert dt <= 1e-8 and ang <= 1e-9, (dt, ang)
    for name, method in (("lum", wm.WM_INFO_LUM), ("censi", wm.WM_INFO_CENSI), ("lumold", wm.WM_INFO_LUMOLD)):
        rc, info, deg = m.icp_info(method, got["T"], lin_covar=2.5e-4, ang_covar=7.78e-9, max_corr=3.0)
        rc1, info1, deg1 = infos[name]
        assert rc == rc1 == 0 and bool(deg) == bool(deg1)
        assert np.allclose(info, info1, rtol=2e-5, atol=1e-9 * np.abs(info1).max()), (name, np.abs(info - info1).max())
    m.close()
