"""Many small registrations per launch (wm_icp_batch_match, csrc/wm_small.hip): every item must
equal what the reference does for that pair -- ICPMatcher::match()'s full-resolution branch
(wave_matching/src/icp.cpp:123-131) followed by estimateInfo() (icp.cpp:135-142, whose result is
estimateLUMold's, icp_pcl_functions.cpp:51-179) -- as restated by the oracle, and must equal the
one-pair-at-a-time device path."""
import numpy as np
import pytest
import torch  # (before the HIP library is loaded: see test_fullsize_gpu.py)

from helpers import pose_error
from libwave_amd import synth

pytestmark = pytest.mark.gpu


def _pairs(sizes, seed0=100):
    out = []
    for k, n in enumerate(sizes):
        ref, tgt, T_gt = synth.pair(n, seed=seed0 + k, mode="resample")
        out.append((ref, tgt, T_gt))
    return out


def test_batch_items_match_the_oracle_and_the_single_path(wm, ctx, oracle):
    """BASELINE configs[0]-sized pairs (10 000 points) and smaller ones, default stopping rules."""
    pairs = _pairs([10000, 10000, 7000, 2500, 9999, 64])
    got = ctx.icp_batch_match([(r, t) for r, t, _ in pairs], with_info=True, max_corr=3.0, max_iter=100)
    assert len(got) == len(pairs)
    for (ref, tgt, T_gt), g in zip(pairs, got):
        want = oracle.IcpMatch(ref, tgt, res=-1.0, multiscale_steps=0, incremental_float=0)
        assert g["rc"] == 0 and want.ok and g["converged"]
        assert (g["iterations"], g["state"]) == (want.r.iterations, wm.CONV_NAMES.get(want.r.state, want.r.state))
        assert g["n_corr"] == want.r.n_corr
        dt, ang = pose_error(g["T"], want.T)
        assert dt <= 1e-7 and ang <= 1e-8, (dt, ang)
        olumold, _ = want.lumold(3.0)
        np.testing.assert_allclose(g["info"], olumold, rtol=2e-4, atol=1e-6 * np.abs(olumold).max())
        # the one-pair device path: same correspondences, sums in another order
        one = ctx.icp_match(ref, tgt, res=-1.0, max_corr=3.0, max_iter=100, carry_state=0)
        assert (one["iterations"], one["state"], one["n_corr"]) == (g["iterations"], g["state"], g["n_corr"])
        dt, ang = pose_error(g["T"], one["T"])
        assert dt <= 1e-9 and ang <= 1e-10, (dt, ang)
        rc, lumold, _ = ctx.icp_info(wm.WM_INFO_LUMOLD, max_corr=3.0)
        np.testing.assert_allclose(g["info"], lumold, rtol=1e-5, atol=1e-9 * np.abs(lumold).max())


@pytest.mark.parametrize("mode", ["svd", "gn6"])
def test_batch_forced_iterations(wm, ctx, oracle, mode):
    m = wm.WM_ICP_SVD if mode == "svd" else wm.WM_ICP_GN6
    pairs = _pairs([10000, 4000, 9000], seed0=300)
    got = ctx.icp_batch_match([(r, t) for r, t, _ in pairs], with_info=False, max_corr=3.0, force_iterations=20, mode=m)
    for (ref, tgt, _), g in zip(pairs, got):
        want = oracle.icp_align(ref, tgt, max_corr=3.0, force_iterations=20, incremental_float=0,
                                mode=1 if mode == "gn6" else 0)
        assert g["rc"] == 0 and g["state"] == "FORCED" and g["iterations"] == 20
        assert g["n_corr"] == want["n_corr"]
        dt, ang = pose_error(g["T"], want["T"])
        assert dt <= 1e-7 and ang <= 1e-8, (dt, ang)


def test_batch_edge_cases(wm, ctx, oracle):
    rng = np.random.default_rng(7)
    ref, tgt, _ = synth.pair(3000, seed=5, mode="resample")
    far = (tgt + np.array([500.0, 0, 0], np.float32)).astype(np.float32)       # nothing within max_corr
    holes = ref.copy()
    holes[::7, 1] = np.nan                                                      # non-finite points are dropped
    holes_t = tgt.copy()
    holes_t[::5, 0] = np.inf
    flat = np.zeros((500, 3), np.float32)
    flat[:, :2] = rng.uniform(-5, 5, (500, 2))                                  # a plane: one layer of cells
    same = np.tile(np.array([[1.0, 2.0, 3.0]], np.float32), (100, 1))           # all points identical
    empty = np.zeros((0, 3), np.float32)
    pairs = [(ref, far), (holes, holes_t), (flat, flat + np.float32(0.01)), (same, same), (empty, tgt), (ref, empty),
             (ref[:2], tgt)]
    got = ctx.icp_batch_match(pairs, with_info=True, max_corr=3.0, max_iter=50)
    assert got[0]["rc"] == wm.WM_TOO_FEW and got[0]["state"] == "NO_CORRESPONDENCES"
    keep_r, keep_t = np.isfinite(holes).all(1), np.isfinite(holes_t).all(1)
    want = oracle.icp_align(holes[keep_r], holes_t[keep_t], max_corr=3.0, max_iter=50, incremental_float=0)
    assert got[1]["rc"] == 0 and (got[1]["iterations"], got[1]["n_corr"]) == (want["iterations"], want["n_corr"])
    dt, ang = pose_error(got[1]["T"], want["T"])
    assert dt <= 1e-7 and ang <= 1e-8
    want = oracle.icp_align(flat, flat + np.float32(0.01), max_corr=3.0, max_iter=50, incremental_float=0)
    assert got[2]["rc"] == 0 and got[2]["iterations"] == want["iterations"]
    dt, ang = pose_error(got[2]["T"], want["T"])
    assert dt <= 1e-6 and ang <= 1e-6
    assert got[3]["rc"] in (0, wm.WM_NOT_CONVERGED)                             # degenerate: must not hang
    assert got[4]["rc"] == wm.WM_TOO_FEW and got[5]["rc"] == wm.WM_TOO_FEW
    assert got[6]["rc"] == wm.WM_TOO_FEW                        # < 3 correspondences (PCL)


def test_batch_source_larger_than_the_sort_window(wm, ctx, oracle):
    """Sources beyond 16 384 points keep the caller's order (only the TARGET has to fit the LDS)."""
    ref, _, _ = synth.pair(20000, seed=11, mode="resample")
    _, tgt, _ = synth.pair(9000, seed=11, mode="resample")
    holes = ref.copy()
    holes[::9, 2] = np.nan
    got = ctx.icp_batch_match([(ref, tgt), (holes, tgt)], with_info=True, max_corr=3.0, max_iter=60)
    for cloud, g in zip((ref, holes[np.isfinite(holes).all(1)]), got):
        want = oracle.IcpMatch(cloud, tgt, res=-1.0, multiscale_steps=0, incremental_float=0, max_corr=3.0, max_iter=60)
        assert g["rc"] == 0 and (g["iterations"], g["n_corr"]) == (want.r.iterations, want.r.n_corr)
        dt, ang = pose_error(g["T"], want.T)
        assert dt <= 1e-7 and ang <= 1e-8, (dt, ang)
        olumold, _ = want.lumold(3.0)
        np.testing.assert_allclose(g["info"], olumold, rtol=2e-4, atol=1e-6 * np.abs(olumold).max())


def test_batch_targets_beyond_the_lds_stay_in_hbm(wm, ctx, oracle, testscan):
    """Targets of 10 001 ... 65 535 points: the same kernel with the cell-sorted target in HBM scratch.
    Mixed with LDS-resident items in one call; the reference's own scan (55 067 points) among them."""
    a = synth.pair(30000, seed=21, mode="resample")
    b = synth.pair(12000, seed=22, mode="resample")
    c = synth.pair(8000, seed=23, mode="resample")
    perturb = np.eye(4)
    perturb[0, 3] = 0.2
    scan_t = oracle.transform_cloud_d(testscan, perturb)
    holes = a[1].copy()
    holes[::11, 0] = np.nan
    pairs = [(a[0], a[1]), (c[0], c[1]), (b[0], b[1]), (testscan, scan_t), (a[0], holes)]
    got = ctx.icp_batch_match(pairs, with_info=True, max_corr=3.0, max_iter=100)
    for (ref, tgt), g in zip(pairs, got):
        keep = np.isfinite(tgt).all(1)
        want = oracle.IcpMatch(ref, tgt[keep], res=-1.0, multiscale_steps=0, incremental_float=0)
        assert g["rc"] == 0 and want.ok
        assert (g["iterations"], g["n_corr"]) == (want.r.iterations, want.r.n_corr)
        dt, ang = pose_error(g["T"], want.T)
        assert dt <= 1e-6 and ang <= 1e-7, (dt, ang)
        olumold, _ = want.lumold(3.0)
        np.testing.assert_allclose(g["info"], olumold, rtol=2e-4, atol=1e-6 * np.abs(olumold).max())
    assert np.linalg.norm(got[3]["T"] - perturb) < 0.1     # wave_matching/tests/icp_tests.cpp:59-61


def test_batch_rejects_targets_beyond_sixteen_bit_slots(wm, ctx):
    ref, tgt, _ = synth.pair(wm.WM_BATCH_MAX_TARGET_POINTS + 1, seed=1, mode="resample")
    with pytest.raises(wm.WmError):
        ctx.icp_batch_match([(ref, tgt)], max_corr=3.0)


def test_batch_of_many_fills_the_device(wm, ctx, oracle):
    """300 pairs (more workgroups than CUs), device-resident clouds; spot-checked against the oracle."""
    pairs = _pairs([4096] * 6, seed0=900)
    dev = [(torch.from_numpy(r).cuda(), torch.from_numpy(t).cuda()) for r, t, _ in pairs]
    got = ctx.icp_batch_match([dev[k % 6] for k in range(300)], with_info=True, max_corr=3.0, max_iter=100)
    for k in range(300):
        assert got[k]["rc"] == 0
        assert np.array_equal(got[k]["T"], got[k % 6]["T"]) and np.array_equal(got[k]["info"], got[k % 6]["info"])
    for k in (0, 5):
        ref, tgt, _ = pairs[k]
        want = oracle.icp_align(ref, tgt, max_corr=3.0, max_iter=100, incremental_float=0)
        dt, ang = pose_error(got[k]["T"], want["T"])
        assert dt <= 1e-7 and ang <= 1e-8


# ---------------------------------------------------------------- the voxel-filtered branches, batched
def _scan_pairs(oracle, testscan):
    shift = np.eye(4)
    shift[0, 3] = 0.2
    rng = np.random.default_rng(5)
    jit = (oracle.transform_cloud_d(testscan, shift) + rng.uniform(-0.3, 0.3, testscan.shape)).astype(np.float32)
    a = synth.pair(30000, seed=41, mode="resample")
    b = synth.pair(9000, seed=42, mode="resample")
    holes = a[0].copy()
    holes[::13, 2] = np.inf
    return [(testscan, oracle.transform_cloud_d(testscan, shift)), (a[0], a[1]), (testscan, jit), (b[0], b[1]), (holes, a[1]),
            (testscan, testscan)]


@pytest.mark.parametrize("leaf", [0.1, 0.4, 1.6])
def test_voxel_grid_of_a_whole_batch_is_bit_exact(wm, ctx, oracle, testscan, leaf):
    """pcl::VoxelGrid of twelve clouds in one pass (one sort key = cloud number above leaf index) against
    the one-cloud device path, which the oracle pins bit for bit (tests/test_match_gpu.py)."""
    pairs = _scan_pairs(oracle, testscan)
    got = ctx.voxel_downsample_batch(pairs, leaf)
    for (ref, tgt), (fr, ft) in zip(pairs, got):
        assert np.array_equal(fr, ctx.voxel_downsample(ref, leaf))
        assert np.array_equal(ft, ctx.voxel_downsample(tgt, leaf))
    assert np.array_equal(got[0][0], oracle.voxel_grid(testscan, leaf))


@pytest.mark.parametrize("res,steps", [(0.1, 0), (0.1, 3), (0.05, 1)])
def test_batched_filtered_matches_equal_one_by_one(wm, ctx, oracle, testscan, res, steps):
    """ICPMatcher::match() with a voxel filter (single scale icp.cpp:105-122, multiscale :77-104) +
    estimateInfo, six pairs per call: every item as wm_icp_match + wm_icp_info give it, and as the oracle does."""
    pairs = _scan_pairs(oracle, testscan)
    got = ctx.icp_batch_match(pairs, with_info=True, res=res, multiscale_steps=steps, max_corr=3.0, max_iter=100)
    for k, ((ref, tgt), g) in enumerate(zip(pairs, got)):
        one = ctx.icp_match(ref, tgt, res=res, multiscale_steps=steps, max_corr=3.0, max_iter=100, carry_state=0)
        assert g["rc"] == one["rc"] == 0, (k, g["rc"], one["rc"])
        assert (g["iterations"], g["state"], g["n_corr"]) == (one["iterations"], one["state"], one["n_corr"]), k
        dt, ang = pose_error(g["T"], one["T"])
        assert dt <= 1e-7 and ang <= 1e-8, (k, dt, ang)
        rc, lumold, _ = ctx.icp_info(wm.WM_INFO_LUMOLD, max_corr=3.0)
        if np.isfinite(lumold).all() and np.abs(lumold).max() < 1e15:      # (pair 5 is an exact copy: s^2 = 0)
            np.testing.assert_allclose(g["info"], lumold, rtol=1e-4, atol=1e-8 * np.abs(lumold).max())
    for k in (0, 2):
        ref, tgt = pairs[k]
        want = oracle.IcpMatch(ref, tgt, res=res, multiscale_steps=steps, incremental_float=0)
        assert want.ok and got[k]["iterations"] == want.r.iterations and got[k]["n_corr"] == want.r.n_corr
        dt, ang = pose_error(got[k]["T"], want.T)
        assert dt <= 1e-6 and ang <= 1e-7, (dt, ang)
        olumold, _ = want.lumold(3.0)
        np.testing.assert_allclose(got[k]["info"], olumold, rtol=2e-4, atol=1e-6 * np.abs(olumold).max())


def test_batched_filtered_match_fails_fast_like_match(wm, ctx, oracle, testscan):
    """A pair with nothing within max_corr stops at the first scale (icp.cpp:96-98); the others go on."""
    far = (testscan + np.array([800.0, 0, 0], np.float32)).astype(np.float32)
    empty = np.zeros((0, 3), np.float32)
    pairs = [(testscan, far), (testscan, testscan), (empty, testscan), (testscan, empty)]
    got = ctx.icp_batch_match(pairs, with_info=True, res=0.1, multiscale_steps=3, max_corr=3.0, max_iter=100)
    one = ctx.icp_match(testscan, far, res=0.1, multiscale_steps=3, max_corr=3.0, max_iter=100, carry_state=0)
    assert got[0]["rc"] == one["rc"] == wm.WM_TOO_FEW and got[0]["T"] is None
    assert got[1]["rc"] == 0 and np.linalg.norm(got[1]["T"] - np.eye(4)) < 1e-6
    assert got[2]["rc"] == wm.WM_TOO_FEW and got[3]["rc"] == wm.WM_TOO_FEW


def test_random_batches_equal_one_by_one(wm, ctx):
    """Forty random pairs per call -- sizes from 3 to 6 000 points, clustered and flat clouds, NaN holes,
    big offsets -- full resolution and voxel-filtered: every item as the one-pair device path gives it."""
    rng = np.random.default_rng(2024)

    def cloud(n):
        kind = rng.integers(0, 4)
        if kind == 0:
            c = rng.uniform(-10, 10, (n, 3))
        elif kind == 1:
            c = np.c_[rng.uniform(-20, 20, (n, 2)), rng.normal(0, 0.02, n)]
        elif kind == 2:
            c = rng.normal(0, 1.0, (n, 3)) * np.array([8.0, 3.0, 0.5]) + np.array([100.0, -50.0, 5.0])
        else:
            c = np.r_[rng.normal(0, 0.3, (n - n // 10, 3)), rng.uniform(-80, 80, (n // 10, 3))]   # a dense core + far outliers
        return c.astype(np.float32)

    pairs = []
    for _ in range(40):
        n = int(rng.integers(3, 6000))
        ref = cloud(n)
        ang = rng.uniform(-0.03, 0.03)
        R = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]], np.float32)
        m = int(rng.integers(3, 6000))
        sel = rng.integers(0, n, m)
        tgt = (ref[sel] @ R.T + rng.uniform(-0.2, 0.2, 3).astype(np.float32) + rng.normal(0, 0.01, (m, 3)).astype(np.float32)).astype(np.float32)
        if rng.random() < 0.3:
            ref[rng.integers(0, n, max(1, n // 20)), rng.integers(0, 3)] = np.nan
        pairs.append((ref, tgt))
    for res, steps in ((-1.0, 0), (0.25, 0), (0.2, 2)):
        got = ctx.icp_batch_match(pairs, with_info=True, res=res, multiscale_steps=steps, max_corr=3.0, max_iter=60)
        for k, ((ref, tgt), g) in enumerate(zip(pairs, got)):
            # a matcher of its own per pair, as a batch item is: PCL's criteria start fresh and keep the
            # last MSE from scale to scale (carry_state = 1 on a new context)
            fresh = wm.Context(0)
            one = fresh.icp_match(ref, tgt, res=res, multiscale_steps=steps, max_corr=3.0, max_iter=60, carry_state=1)
            fresh.close()
            assert g["rc"] == one["rc"], (res, steps, k, g["rc"], one["rc"])
            if one["rc"] != 0:
                continue
            assert (g["iterations"], g["n_corr"]) == (one["iterations"], one["n_corr"]), (res, steps, k)
            dt, ang = pose_error(g["T"], one["T"])
            assert dt <= 1e-6 and ang <= 1e-7, (res, steps, k, dt, ang)


def test_batched_filtered_match_hands_oversized_pairs_to_the_one_pair_path(wm, ctx):
    """A pair whose FILTERED target still has more than 65 535 points is registered by wm_icp_match inside
    the call; its neighbours in the batch are not affected."""
    big = synth.pair(200000, seed=3, mode="resample")
    small = synth.pair(20000, seed=4, mode="resample")
    assert len(ctx.voxel_downsample(big[1], 0.05)) > wm.WM_BATCH_MAX_TARGET_POINTS
    pairs = [(small[0], small[1]), (big[0], big[1]), (small[1], small[0])]
    got = ctx.icp_batch_match(pairs, with_info=True, res=0.05, multiscale_steps=1, max_corr=3.0, max_iter=60)
    for k, ((ref, tgt), g) in enumerate(zip(pairs, got)):
        fresh = wm.Context(0)
        one = fresh.icp_match(ref, tgt, res=0.05, multiscale_steps=1, max_corr=3.0, max_iter=60, carry_state=1)
        rc, lumold, _ = fresh.icp_info(wm.WM_INFO_LUMOLD, max_corr=3.0)
        fresh.close()
        assert g["rc"] == one["rc"] == 0, k
        assert (g["iterations"], g["n_corr"]) == (one["iterations"], one["n_corr"]), k
        dt, ang = pose_error(g["T"], one["T"])
        assert dt <= 1e-6 and ang <= 1e-7, (k, dt, ang)
        np.testing.assert_allclose(g["info"], lumold, rtol=1e-4, atol=1e-8 * np.abs(lumold).max())


def test_batched_matches_do_not_depend_on_where_the_clouds_live(wm, ctx, oracle, testscan):
    """Host clouds (staged through pinned memory) and device-resident clouds: the same bits."""
    pairs = _scan_pairs(oracle, testscan)[:4]
    dev = [(torch.from_numpy(r).cuda(), torch.from_numpy(t).cuda()) for r, t in pairs]
    for kw in (dict(res=-1.0, multiscale_steps=0), dict(res=0.1, multiscale_steps=3)):
        a = ctx.icp_batch_match(pairs[1:], with_info=True, max_corr=3.0, max_iter=100, **kw) if kw["res"] < 0 else \
            ctx.icp_batch_match(pairs, with_info=True, max_corr=3.0, max_iter=100, **kw)
        b = ctx.icp_batch_match(dev[1:], with_info=True, max_corr=3.0, max_iter=100, **kw) if kw["res"] < 0 else \
            ctx.icp_batch_match(dev, with_info=True, max_corr=3.0, max_iter=100, **kw)
        for x, y in zip(a, b):
            assert x["rc"] == y["rc"] == 0 and x["iterations"] == y["iterations"]
            assert np.array_equal(x["T"], y["T"]) and np.array_equal(x["info"], y["info"])


@pytest.mark.parametrize("res,steps", [(-1.0, 0), (0.1, 0), (0.1, 3)])
def test_batch_of_empty_pairs_only(wm, ctx, res, steps):
    """The C API's contract is "WM_OK when the batch ran, one status per item": a batch whose items are
    all empty ran (PCL: empty input -> "Not enough correspondences"), with or without a voxel filter."""
    empty = np.zeros((0, 3), np.float32)
    got = ctx.icp_batch_match([(empty, empty)] * 3, with_info=True, res=res, multiscale_steps=steps, max_corr=3.0)
    assert len(got) == 3
    for g in got:
        assert g["rc"] == wm.WM_ERR_STATE and g["T"] is None and g["state"] == "NO_CORRESPONDENCES"


def test_batch_fallback_pair_carries_the_mse_across_its_scales(wm, ctx):
    """A pair too large for the resident kernel is registered inside the batch by the one-pair path; it
    must stop where a fresh one-pair matcher stops (fresh criteria per pair, MSE carried across scales)."""
    big_r, big_t, _ = synth.pair(150000, seed=61, mode="resample")
    small_r, small_t, _ = synth.pair(9000, seed=62, mode="resample")
    kw = dict(max_corr=3.0, max_iter=100)
    got = ctx.icp_batch_match([(small_r, small_t), (big_r, big_t)], with_info=False, res=0.02, multiscale_steps=2, **kw)
    c2 = wm.Context(0)
    one = c2.icp_match(big_r, big_t, res=0.02, multiscale_steps=2, carry_state=1, **kw)
    c2.close()
    assert got[1]["rc"] == one["rc"] == 0
    assert (got[1]["iterations"], got[1]["state"]) == (one["iterations"], one["state"])
    dt, ang = pose_error(got[1]["T"], one["T"])
    assert dt <= 1e-9 and ang <= 1e-10
