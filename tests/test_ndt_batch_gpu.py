"""Many NDT registrations per launch (wm_ndt_batch_match, csrc/wm_ndt_small.hip: one pair per compute unit, the
target's voxel model and the whole of pcl::NormalDistributionsTransform::align inside the kernel -- what a
wave::MultiMatcher<NDTMatcher>, wave_matching/include/wave/matching/multi_matcher.hpp:29-34, has waiting in its queue).
Every item must be what NDTMatcher::match() gives for that pair (wave_matching/src/ndt.cpp:48-65) -- as restated by
the oracle, and as the one-pair device path computes it.  Voxel membership, radius tests, the terms of the derivative
passes and the control's code (wm_ndt_ctl.hpp, compiled for both sides) are the same; a voxel's sums are formed in
double-double instead of in point order, the passes' sums in another order, exp / log / sin / cos by the device
library: the bar is 1e-6 m / 1e-6 rad against the one-pair path with the same iteration count (on most of these
inputs the transforms come out EQUAL), north_star's 1e-4 against the oracle."""
import numpy as np
import pytest
import torch  # (before the HIP library is loaded: see test_fullsize_gpu.py)

from helpers import TOL_R, TOL_T, pose_error
from libwave_amd import synth

pytestmark = pytest.mark.gpu


def _one(ctx, ref, tgt, **kw):
    ctx.set_source(ref)
    ctx.set_target(tgt)
    return ctx.ndt_align(**kw)


def _close(g, one, tol_t=1e-6, tol_r=1e-6):
    assert g["rc"] == one["rc"], (g, one)
    assert (g["converged"], g["n_voxels"]) == (one["converged"], one["n_voxels"])
    if g["T"] is None or one["T"] is None:
        assert g["T"] is None and one["T"] is None
        return
    # (the same Newton iterations; the number of PASSES may differ: a line search on the flat bottom of the objective
    # takes two trials or ten depending on the sixteenth digit, and ends at the same step either way)
    assert g["iterations"] == one["iterations"], (g, one)
    dt, ang = pose_error(g["T"], one["T"])
    assert dt <= tol_t and ang <= tol_r, (dt, ang)
    assert abs(g["score"] - one["score"]) <= 1e-9 * max(abs(one["score"]), 1e-30)


def test_batch_items_equal_the_one_pair_path_and_the_oracle(wm, ctx, oracle):
    sizes = [20000, 12000, 30000, 8000, 20001]
    pairs = [synth.pair(n, seed=700 + k, mode="resample") for k, n in enumerate(sizes)]
    for res, kw in ((1.0, dict(step_size=0.1, t_eps=1e-6)), (2.5, dict()), (5.0, dict()), (1.0, dict(step_size=0.1, skip_line_search=1, max_iter=30))):
        got = ctx.ndt_batch_match([(r, t) for r, t, _ in pairs], res=res, **kw)
        for (ref, tgt, T_gt), g in zip(pairs, got):
            _close(g, _one(ctx, ref, tgt, res=res, **kw))
    # against the oracle (a size it finishes in seconds)
    ref, tgt, _ = pairs[3]
    g = ctx.ndt_batch_match([(ref, tgt)], res=1.0, step_size=0.1, t_eps=1e-6)[0]
    want = oracle.ndt_align(ref, tgt, res=1.0, step_size=0.1, t_eps=1e-6)
    assert g["rc"] == 0 and want["converged"] and g["n_voxels"] == want["n_voxels"] and g["iterations"] == want["iterations"]
    dt, ang = pose_error(g["T"], want["T"])
    assert dt <= TOL_T and ang <= TOL_R, (dt, ang)


def test_reference_scan_through_the_batch(wm, ctx, oracle, testscan):
    """testscan.pcd against itself shifted by 0.2 m (wave_matching/tests/ndt_tests.cpp:86-102) at voxel sizes whose
    lattice fits the kernel's table, and at the reference's 0.3 m, whose lattice does not: that pair is registered
    by the one-pair path inside the same call."""
    P = np.eye(4)
    P[0, 3] = 0.2
    target = oracle.transform_cloud_d(testscan, P)
    for res in (2.0, 5.0, 0.3):
        g = ctx.ndt_batch_match([(testscan, target), (testscan, testscan)], res=res, step_size=3, max_iter=100, t_eps=1e-8)
        one = _one(ctx, testscan, target, res=res, step_size=3, max_iter=100, t_eps=1e-8)
        _close(g[0], one)
        _close(g[1], _one(ctx, testscan, testscan, res=res, step_size=3, max_iter=100, t_eps=1e-8))
        if res == 0.3:
            assert g[0]["rc"] == 0 and np.linalg.norm(g[0]["T"] - P) < 0.12  # ndt_tests.cpp:37 threshold


def test_batch_edge_cases(wm, ctx):
    ref, tgt, _ = synth.pair(15000, seed=21, mode="resample")
    empty = np.zeros((0, 3), np.float32)
    nan_ref = ref.copy()
    nan_ref[::13] = np.nan
    nan_tgt = tgt.copy()
    nan_tgt[7::29, 2] = np.inf
    far = tgt + np.float32(900.0)  # no voxel anywhere near the source
    few = tgt[:40]                 # hardly a voxel with six points
    pairs = [(ref, tgt), (empty, tgt), (ref, empty), (nan_ref, nan_tgt), (ref, far), (ref, few), (empty, empty), (ref[:3], tgt)]
    got = ctx.ndt_batch_match(pairs, res=1.0)
    assert [got[k]["rc"] for k in (1, 2, 6)] == [wm.WM_ERR_STATE] * 3
    for k in (0, 3, 4, 5, 7):
        _close(got[k], _one(ctx, *pairs[k], res=1.0))
    # an item's result does not depend on its neighbours in the batch, nor on the order; and the call is repeatable
    again = ctx.ndt_batch_match([pairs[3], pairs[0]], res=1.0)
    assert np.array_equal(again[0]["T"], got[3]["T"]) and np.array_equal(again[1]["T"], got[0]["T"])
    assert ctx.ndt_batch_match([], res=1.0) == []
    # one far-away FINITE outlier (a garbage lidar return): |coordinate / res| >= 2^31 must not overflow the lattice
    # arithmetic of the batched kernel (round 4's advisor finding) -- the item is handed to the one-pair path, which
    # refuses a lattice of more than 2^20 voxels along an axis; its neighbours in the batch are registered as ever
    out_tgt = tgt.copy()
    out_tgt[5] = (np.float32(3.0e9), np.float32(-2.5e9), np.float32(1.0))
    out_ref = ref.copy()
    out_ref[9, 0] = np.float32(-4.0e9)
    got2 = ctx.ndt_batch_match([(ref, out_tgt), (ref, tgt), (out_ref, tgt)], res=1.0)
    with pytest.raises(wm.WmError):  # (the one-pair call raises for its WM_ERR_ARG)
        _one(ctx, ref, out_tgt, res=1.0)
    assert got2[0]["rc"] == wm.WM_ERR_ARG and got2[0]["T"] is None
    assert np.array_equal(got2[1]["T"], got[0]["T"])
    _close(got2[2], _one(ctx, out_ref, tgt, res=1.0))
    with pytest.raises(wm.WmError):
        ctx.ndt_batch_match([(ref, tgt)], res=0.0)


def test_batch_device_clouds_and_many_pairs(wm, ctx):
    base = [synth.pair(n, seed=900 + k, mode="resample") for k, n in enumerate((6000, 9000, 7000))]
    first = [_one(ctx, r, t, res=2.0) for r, t, _ in base]
    dev = []
    for r, t, _ in base:
        r4 = np.zeros((len(r), 4), np.float32)
        t4 = np.zeros((len(t), 4), np.float32)
        r4[:, :3], t4[:, :3] = r, t
        r4[:, 3], t4[:, 3] = 3.0, -2.0
        dev.append((torch.from_numpy(r4).cuda(), torch.from_numpy(t4).cuda()))
    for g, w in zip(ctx.ndt_batch_match(dev, res=2.0), first):
        _close(g, w)
    # 300 pairs in one call (more workgroups than compute units)
    got = ctx.ndt_batch_match([(base[k % 3][0], base[k % 3][1]) for k in range(300)], res=2.0)
    for k, g in enumerate(got):
        _close(g, first[k % 3])
        assert np.array_equal(g["T"], got[k % 3]["T"])


@pytest.mark.parametrize("kw", [dict(pcl_d1_sign=0), dict(force_iterations=6), dict(outlier_ratio=0.3), dict(step_size=0.05, max_iter=8)])
def test_batch_follows_the_parameters(wm, ctx, kw):
    """the true second derivative instead of PCL's sign slip, forced iterations (the bench mode), another outlier ratio,
    a registration cut short by max_iter: the same as the one-pair path each time"""
    pairs = [synth.pair(n, seed=1100 + k, mode="resample") for k, n in enumerate((14000, 9000))]
    got = ctx.ndt_batch_match([(r, t) for r, t, _ in pairs], res=1.5, **kw)
    for (r, t, _), g in zip(pairs, got):
        _close(g, _one(ctx, r, t, res=1.5, **kw))
    if "force_iterations" in kw:
        assert all(g["iterations"] == 6 for g in got)
