"""Many GICP registrations per launch (wm_gicp_batch_match, csrc/wm_gicp_small.hip: one pair per compute unit, the
whole of pcl::GeneralizedIterativeClosestPoint::align inside the kernel -- what a wave::MultiMatcher<GICPMatcher>,
wave_matching/include/wave/matching/multi_matcher.hpp:29-34, has waiting in its queue).  Every item must be what
GICPMatcher::match() gives for that pair (wave_matching/src/gicp.cpp:37-64) -- as restated by the oracle, and as the
one-pair device path computes it.  The neighbours, covariances, Mahalanobis matrices and the objective's
double-double sums are the same code on both device paths; the optimiser's scalar code is compiled for the device
here, with glibc's sinf / cosf / atan2f / asinf restated (tests/test_bfgs_trig_cpu.py) so that the float transform
of every evaluation has the same bits: on these inputs the results are EQUAL, not close.  (The contract in
include/wavematch.h only promises ~1e-6 m: one sinf argument in ten million rounds differently, and the DOUBLE
sin / cos in the gradient's rotation part are the device library's -- a last-bit difference there showed after 300
evaluations of a hopeless 64-point registration that runs into max_iter with f ~ 4e4, scripts/dev/dev_gicp_batch_trace.py.)"""
import numpy as np
import pytest
import torch  # (before the HIP library is loaded: see test_fullsize_gpu.py)

from libwave_amd import synth

pytestmark = pytest.mark.gpu

KEYS = ("rc", "converged", "iterations", "inner_total", "evaluations", "n_corr")

# both forms of the objective (include/wavematch.h: wm_gicp_params::objective), each against the oracle's restatement
# of it (oracle/gicp.c: wmo_gicp_set_objective) and against the one-pair path in the same form
OBJECTIVES = [("statistics", 1, 1), ("pcl_sums", 0, 0)]   # (name, wm_gicp_params::objective, oracle objective mode)


@pytest.fixture(params=OBJECTIVES, ids=[o[0] for o in OBJECTIVES])
def objective(request, oracle):
    name, hip, orc = request.param
    oracle.gicp_set_objective(orc)
    yield hip
    oracle.gicp_set_objective(0)


def _pairs(sizes, seed0=100, mode="resample"):
    return [synth.pair(n, seed=seed0 + k, mode=mode) for k, n in enumerate(sizes)]


def _same(a, b):
    assert tuple(a[k] for k in KEYS) == tuple(b[k] for k in KEYS), (a, b)
    if a["T"] is None or b["T"] is None:
        assert a["T"] is None and b["T"] is None
    else:
        assert np.array_equal(a["T"], b["T"]), np.abs(a["T"] - b["T"]).max()
    assert a["f"] == b["f"]


def test_batch_items_equal_the_oracle_and_the_one_pair_path(wm, ctx, oracle, objective):
    pairs = _pairs([3000, 1200, 5000, 400, 2999, 8000])
    got = ctx.gicp_batch_match([(r, t) for r, t, _ in pairs], objective=objective)
    assert len(got) == len(pairs)
    for (ref, tgt, T_gt), g in zip(pairs, got):
        one = ctx.gicp_match(ref, tgt, objective=objective)
        _same(g, one)
        assert g["rc"] == 0
        if len(ref) >= 5000:  # (sparser samplings of this 100 x 60 m scene do not pin the pose)
            assert np.linalg.norm(g["T"][:3, 3] - T_gt[:3, 3]) < 0.05
        if len(ref) <= 3000:
            want = oracle.gicp_align(ref, tgt)
            assert want["converged"] and g["n_corr"] == want["n_corr"]
            assert (g["iterations"], g["inner_total"]) == (want["iterations"], want["inner_total"])
            assert np.array_equal(g["T"], want["T"])


CASES = [("fullResNullMatch", -1.0, 0.0), ("nullDisplacement", 0.05, 0.0), ("smallDisplacement", 0.05, 0.2)]


def test_reference_gicp_cases_through_the_batch(wm, ctx, oracle, testscan, objective):
    """wave_matching/tests/gicp_tests.cpp's three cases on testscan.pcd, queued together (the voxel-filtered two in
    one call: res is the matcher's parameter, not the pair's)."""
    def target_of(tx):
        P = np.eye(4)
        P[0, 3] = tx
        return oracle.transform_cloud_d(testscan, P), P
    for res in (-1.0, 0.05):
        cases = [c for c in CASES if c[1] == res]
        pairs = [(testscan, target_of(tx)[0]) for _, _, tx in cases]
        got = ctx.gicp_batch_match(pairs, res=res, objective=objective)
        for (name, _, tx), (ref, tgt), g in zip(cases, pairs, got):
            P = target_of(tx)[1]
            assert g["rc"] == 0 and g["converged"], name
            assert np.linalg.norm(g["T"] - P) < 0.1  # gicp_tests.cpp:36 threshold
            _same(g, ctx.gicp_match(ref, tgt, res=res, objective=objective))
            a = ref if res < 0 else oracle.voxel_grid(ref, res)
            b = tgt if res < 0 else oracle.voxel_grid(tgt, res)
            want = oracle.gicp_align(a, b)
            assert np.array_equal(g["T"], want["T"]) and g["n_corr"] == want["n_corr"], name


def test_batch_edge_cases(wm, ctx, objective):
    ref, tgt, _ = synth.pair(2000, seed=7, mode="resample")
    empty = np.zeros((0, 3), np.float32)
    nan_ref = ref.copy()
    nan_ref[::17] = np.nan
    nan_tgt = tgt.copy()
    nan_tgt[5::23, 1] = np.inf
    far = tgt + np.float32(500.0)  # nothing within max_corr = 5 m
    pairs = [(ref, tgt), (empty, tgt), (ref, empty), (ref[:5], tgt), (ref, tgt[:9]), (nan_ref, nan_tgt), (ref, far), (ref[:10], tgt[:10]),
             (empty, empty)]
    got = ctx.gicp_batch_match(pairs, objective=objective)
    assert [g["rc"] for g in got[1:5]] == [wm.WM_ERR_STATE, wm.WM_ERR_STATE, wm.WM_NOT_CONVERGED, wm.WM_NOT_CONVERGED]
    assert got[6]["rc"] == wm.WM_TOO_FEW and got[6]["T"] is None and got[6]["n_corr"] == 0
    assert got[8]["rc"] == wm.WM_ERR_STATE
    for k in (0, 5, 7):
        _same(got[k], ctx.gicp_match(*pairs[k], objective=objective))
    one_far = ctx.gicp_match(ref, far, objective=objective)
    assert one_far["rc"] == got[6]["rc"]
    # an item's result does not depend on its neighbours in the batch, nor on the order
    again = ctx.gicp_batch_match([pairs[5], pairs[0]], objective=objective)
    _same(again[0], got[5])
    _same(again[1], got[0])
    assert ctx.gicp_batch_match([]) == []


def test_batch_device_clouds_strides_and_parameters(wm, ctx):
    pairs = _pairs([4000, 2500], seed0=300)
    want = [ctx.gicp_match(r, t, corr_rand=15, max_iter=3, r_eps=1e-6) for r, t, _ in pairs]
    # float4 rows in device memory
    dev = []
    for r, t, _ in pairs:
        r4 = np.zeros((len(r), 4), np.float32)
        t4 = np.zeros((len(t), 4), np.float32)
        r4[:, :3], t4[:, :3] = r, t
        r4[:, 3], t4[:, 3] = 7.0, -1.0
        dev.append((torch.from_numpy(r4).cuda(), torch.from_numpy(t4).cuda()))
    got = ctx.gicp_batch_match(dev, corr_rand=15, max_iter=3, r_eps=1e-6)
    for g, w in zip(got, want):
        _same(g, w)
    # the other kernel instantiations of the neighbour list (k = 20, 32)
    for k in (20, 32):
        got = ctx.gicp_batch_match([(r, t) for r, t, _ in pairs], corr_rand=k, max_iter=2)
        for (r, t, _), g in zip(pairs, got):
            _same(g, ctx.gicp_match(r, t, corr_rand=k, max_iter=2))
    # forced iterations (bench mode)
    got = ctx.gicp_batch_match([(r, t) for r, t, _ in pairs], force_iterations=2)
    for (r, t, _), g in zip(pairs, got):
        _same(g, ctx.gicp_match(r, t, force_iterations=2))
        assert g["iterations"] == 2


def test_more_pairs_than_compute_units(wm, ctx):
    """300 small pairs in one call (the 257th workgroup starts when a compute unit frees up)."""
    base = _pairs([600, 450, 800], seed0=500)
    pairs = [(base[k % 3][0], base[k % 3][1]) for k in range(300)]
    got = ctx.gicp_batch_match(pairs)
    first = [ctx.gicp_match(r, t) for r, t, _ in base]
    for k, g in enumerate(got):
        _same(g, first[k % 3])


def test_batch_argument_checks(wm, ctx):
    ref, tgt, _ = synth.pair(600, seed=9, mode="resample")
    with pytest.raises(wm.WmError):
        ctx.gicp_batch_match([(ref, tgt)], corr_rand=0)
    with pytest.raises(wm.WmError):
        ctx.gicp_batch_match([(ref, tgt)], corr_rand=33)
    with pytest.raises(wm.WmError):
        ctx.gicp_batch_match([(ref, tgt)], max_corr=0.0)
    # a cloud beyond the resident kernel's size is refused at full resolution ...
    big = np.zeros((wm.WM_GICP_BATCH_MAX_POINTS + 1, 3), np.float32)
    with pytest.raises(wm.WmError):
        ctx.gicp_batch_match([(big, tgt)])
    # ... and the context is still usable afterwards
    got = ctx.gicp_batch_match([(ref, tgt)])
    assert got[0]["rc"] in (wm.WM_OK, wm.WM_NOT_CONVERGED)
