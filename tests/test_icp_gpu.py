"""GPU parity tests for the ICP hot path: HIP kernels (through the C ABI) vs the
CPU oracle on identical seeded inputs.  Index / distance work is bit-exact;
transforms are within 1e-4 m / 1e-4 rad (north_star tolerance)."""
import numpy as np
import pytest

from helpers import TOL_R, TOL_T, pose_error, svd_stats_numpy
from libwave_amd import synth

pytestmark = pytest.mark.gpu


def _check_nn(wm, ctx, oracle, ref, tgt, T, max_corr, method):
    ctx.set_source(ref)
    ctx.set_target(tgt)
    gi, gd = ctx.nn_search(T, max_corr, method)
    moved = oracle.transform_cloud_f(ref, T.astype(np.float32))
    oi, od = oracle.KdTree(tgt).nn(moved)
    thr = np.float32(max_corr * max_corr)
    keep = od.astype(np.float64) <= float(max_corr) ** 2
    want_idx = np.where(keep, oi, -1)
    assert np.array_equal(gi, want_idx), "match indices differ at %d points" % (gi != want_idx).sum()
    assert np.array_equal(gd[keep], od[keep]), "squared distances are not bit-identical"
    return keep.sum()


@pytest.mark.parametrize("method", ["grid", "brute"])
@pytest.mark.parametrize("max_corr", [3.0, 0.3])
def test_nn_bit_exact_vs_kdtree(wm, ctx, oracle, method, max_corr):
    ref, tgt, _ = synth.pair(20000, seed=11, mode="resample")
    T = synth.make_T((0.3, -0.2, 0.1), (0.02, -0.01, 0.04))
    m = wm.WM_NN_GRID if method == "grid" else wm.WM_NN_BRUTE
    n = _check_nn(wm, ctx, oracle, ref, tgt, T, max_corr, m)
    assert n > 1000


def test_nn_identity_copy_has_zero_distance(wm, ctx, oracle):
    ref = synth.scene(30000, seed=3)
    ctx.set_source(ref)
    ctx.set_target(ref)
    gi, gd = ctx.nn_search(np.eye(4), 3.0, wm.WM_NN_GRID)
    assert np.all(gd == 0)
    # duplicates in the cloud resolve to the lowest index
    oi, _ = oracle.KdTree(ref).nn(ref)
    assert np.array_equal(gi, oi)


def test_nn_far_apart_clouds_have_no_matches(wm, ctx):
    ref = synth.scene(5000, seed=5)
    ctx.set_source(ref + np.float32(500.0))
    ctx.set_target(ref)
    gi, _ = ctx.nn_search(np.eye(4), 3.0, wm.WM_NN_GRID)
    assert np.all(gi == -1)
    r = ctx.icp_align(max_corr=3.0, nn_method=wm.WM_NN_GRID)
    assert r["rc"] == wm.WM_TOO_FEW and r["T"] is None and r["state"] == "NO_CORRESPONDENCES"


def test_nn_partial_overlap_uses_coarse_levels(wm, ctx, oracle):
    """Shifted by 1.5 m: most queries are not certified by the fine level's ring."""
    ref, tgt, _ = synth.pair(20000, seed=21, mode="resample")
    T = synth.make_T((1.5, 0.7, 0.0), (0.0, 0.0, 0.05))
    _check_nn(wm, ctx, oracle, ref, tgt, T, 3.0, wm.WM_NN_GRID)


def test_nn_nonfinite_points_are_dropped(wm, ctx, oracle):
    ref, tgt, _ = synth.pair(8000, seed=13, mode="resample")
    ref = ref.copy()
    tgt = tgt.copy()
    ref[::97, 1] = np.nan
    tgt[::89, 2] = np.inf
    ctx.set_source(ref)
    ctx.set_target(tgt)
    ns, nt = ctx.sizes()
    assert ns == np.isfinite(ref).all(1).sum() and nt == np.isfinite(tgt).all(1).sum()
    for method in (wm.WM_NN_GRID, wm.WM_NN_BRUTE):
        gi, gd = ctx.nn_search(np.eye(4), 3.0, method)
        ok_t = np.isfinite(tgt).all(1)
        tidx = np.nonzero(ok_t)[0]
        oi, od = oracle.KdTree(tgt[ok_t]).nn(np.nan_to_num(ref))
        want = np.where(np.isfinite(ref).all(1) & (od <= 9.0), tidx[oi], -1)
        assert np.array_equal(gi, want)


@pytest.mark.parametrize("mode", ["svd", "gn6"])
def test_stats_reduction_matches_numpy(wm, ctx, mode):
    ref, tgt, _ = synth.pair(50000, seed=17, mode="resample")
    T = synth.make_T((0.1, 0.05, -0.02), (0.0, 0.01, -0.02))
    ctx.set_source(ref)
    ctx.set_target(tgt)
    gi, gd = ctx.nn_search(T, 3.0, wm.WM_NN_GRID)
    keep = gi >= 0
    Tf = T.astype(np.float32)
    x, y, z = ref[:, 0], ref[:, 1], ref[:, 2]
    moved = np.stack([((Tf[r, 0] * x + Tf[r, 1] * y) + Tf[r, 2] * z) + Tf[r, 3] for r in range(3)], 1)
    p, q = moved[keep], tgt[gi[keep]]
    if mode == "svd":
        got = ctx.icp_stats_for(T, wm.WM_ICP_SVD)
        want = svd_stats_numpy(p, q, gd[keep])
        np.testing.assert_allclose(got[:17], want[:17], rtol=1e-12, atol=1e-9)
        rc, Tk = wm.umeyama_from_stats(got)
        assert rc == 0 and abs(np.linalg.det(Tk[:3, :3]) - 1) < 1e-12
    else:
        got = ctx.icp_stats_for(T, wm.WM_ICP_GN6)
        pd, qd = p.astype(np.float64), q.astype(np.float64)
        r = pd - qd
        J = np.zeros((len(pd), 3, 6))
        J[:, 0, 0] = J[:, 1, 1] = J[:, 2, 2] = 1
        J[:, 0, 4], J[:, 0, 5] = pd[:, 2], -pd[:, 1]
        J[:, 1, 3], J[:, 1, 5] = -pd[:, 2], pd[:, 0]
        J[:, 2, 3], J[:, 2, 4] = pd[:, 1], -pd[:, 0]
        H = np.einsum("nca,ncb->ab", J, J)
        g = np.einsum("nca,nc->a", J, r)
        iu = np.triu_indices(6)
        np.testing.assert_allclose(got[2:23], H[iu], rtol=1e-11, atol=1e-7)
        np.testing.assert_allclose(got[23:29], g, rtol=1e-11, atol=1e-7)
        assert got[0] == len(pd)


@pytest.mark.parametrize("n,method", [(10000, "grid"), (10000, "brute"), (100000, "grid")])
def test_icp_forced_iterations_match_oracle(wm, ctx, oracle, n, method):
    """BASELINE config 1 (10k<->10k) and a 100k case at equal iteration count."""
    ref, tgt, T_gt = synth.pair(n, seed=42, mode="resample")
    ctx.set_source(ref)
    ctx.set_target(tgt)
    m = wm.WM_NN_GRID if method == "grid" else wm.WM_NN_BRUTE
    got = ctx.icp_align(max_corr=3.0, force_iterations=20, nn_method=m)
    same = oracle.icp_align(ref, tgt, max_corr=3.0, force_iterations=20, incremental_float=0)
    pcl = oracle.icp_align(ref, tgt, max_corr=3.0, force_iterations=20)  # PCL-literal float path
    assert got["rc"] == 0 and got["state"] == "FORCED" and got["iterations"] == 20
    assert got["n_corr"] == same["n_corr"]
    dt, ang = pose_error(got["T"], same["T"])
    assert dt <= 1e-7 and ang <= 1e-8, (dt, ang)
    dt, ang = pose_error(got["T"], pcl["T"])
    assert dt <= TOL_T and ang <= TOL_R, (dt, ang)
    dt, ang = pose_error(got["T"], T_gt)   # two different samplings of the scene: cm-level
    assert dt < 2e-2 and ang < 2e-3


def test_icp_gn6_reaches_the_same_fixed_point(wm, ctx, oracle):
    ref, tgt, T_gt = synth.pair(20000, seed=42, mode="resample")
    ctx.set_source(ref)
    ctx.set_target(tgt)
    gn = ctx.icp_align(max_corr=3.0, force_iterations=30, mode=wm.WM_ICP_GN6)
    sv = ctx.icp_align(max_corr=3.0, force_iterations=30, mode=wm.WM_ICP_SVD)
    ogn = oracle.icp_align(ref, tgt, max_corr=3.0, force_iterations=30, mode=1, incremental_float=0)
    dt, ang = pose_error(gn["T"], ogn["T"])
    assert dt <= 1e-7 and ang <= 1e-8
    dt, ang = pose_error(gn["T"], sv["T"])
    assert dt <= TOL_T and ang <= TOL_R


def test_icp_free_running_convergence_matches_oracle(wm, ctx, oracle):
    """PCL's stopping rules (TRANSFORM / REL_MSE), tightened as SURVEY 8(d) says."""
    ref, tgt, _ = synth.pair(30000, seed=8, mode="resample")
    ctx.set_source(ref)
    ctx.set_target(tgt)
    for kw in (dict(), dict(t_eps=1e-12, fit_eps=1e-9)):
        got = ctx.icp_align(max_corr=3.0, max_iter=100, carry_state=0, **kw)
        want = oracle.icp_align(ref, tgt, max_corr=3.0, max_iter=100, incremental_float=0, **kw)
        assert got["rc"] == 0 and got["converged"]
        assert (got["iterations"], got["state"]) == (want["iterations"], want["state"])
        dt, ang = pose_error(got["T"], want["T"])
        assert dt <= 1e-7 and ang <= 1e-8


def test_reference_icp_tests_on_testscan(wm, ctx, oracle, testscan):
    """wave_matching/tests/icp_tests.cpp: fullResNullMatch (:45-62) and the full-res
    +0.2 m case; assertion |T - T_gt|_F < 0.1 plus parity with the oracle."""
    for tx in (0.0, 0.2):
        perturb = np.eye(4)
        perturb[0, 3] = tx
        target = oracle.transform_cloud_d(testscan, perturb)
        ctx.set_source(testscan)
        ctx.set_target(target)
        got = ctx.icp_align(max_corr=3.0, max_iter=100, t_eps=1e-8, fit_eps=1e-2, carry_state=0)
        want = oracle.icp_align(testscan, target, incremental_float=0)
        assert got["rc"] == 0
        assert np.linalg.norm(got["T"] - perturb) < 0.1
        assert (got["iterations"], got["state"]) == (want["iterations"], want["state"])
        dt, ang = pose_error(got["T"], want["T"])
        assert dt <= 1e-6 and ang <= 1e-7


def test_correspondences_after_align_match_oracle(wm, ctx, oracle):
    ref, tgt, _ = synth.pair(15000, seed=33, mode="resample")
    ctx.set_source(ref)
    ctx.set_target(tgt)
    got = ctx.icp_align(max_corr=1.0, force_iterations=5)
    want = oracle.icp_align(ref, tgt, max_corr=1.0, force_iterations=5, incremental_float=0,
                            want_corr=True)
    gi, gd = ctx.correspondences()
    assert got["n_corr"] == want["n_corr"]
    mism = (gi != want["corr_idx"]).sum()
    assert mism <= 2, mism   # float T differs in the last bit at most -> rare flips only
    same = gi == want["corr_idx"]
    np.testing.assert_allclose(gd[same & (gi >= 0)], want["corr_d2"][same & (gi >= 0)], rtol=1e-4,
                               atol=1e-9)


def test_empty_and_tiny_inputs(wm, ctx):
    pts = synth.scene(1000, seed=1)
    ctx.set_source(pts[:0])
    ctx.set_target(pts)
    r = ctx.icp_align()
    assert r["rc"] == wm.WM_TOO_FEW
    ctx.set_source(pts[:2])   # fewer than 3 correspondences
    r = ctx.icp_align()
    assert r["rc"] == wm.WM_TOO_FEW and r["state"] == "NO_CORRESPONDENCES"
    ctx.set_source(pts[:3])
    r = ctx.icp_align()
    assert r["rc"] == 0
    assert np.linalg.norm(r["T"] - np.eye(4)) < 1e-6


def test_host_clouds_early_source_path_equals_device_clouds(wm):
    """wm_set_target with a HOST cloud right behind a new source runs the source's bounding box, Morton
    sort and gather under the target's upload (own staging buffer, no drain of the stream).  Same
    registration, bit for bit, as with that overlap off -- with
    non-finite points in both clouds, strides 12 and 16, and clouds re-set in every order."""
    import os
    ref, tgt, _ = synth.pair(40000, seed=21, mode="resample")
    ref = ref.copy()
    tgt = tgt.copy()
    ref[::977] = np.nan
    tgt[5::1013, 1] = np.inf
    out = []
    for early in ("1", "0"):
        os.environ["WM_TUNE_EARLY_SOURCE"] = early
        c = wm.Context(0)
        for _ in range(2):      # twice: the second registration re-uses both staging buffers
            c.set_source(ref)
            c.set_target(tgt)
            r = c.icp_align(max_corr=3.0, force_iterations=15, nn_method=wm.WM_NN_GRID, carry_state=0)
        # a target set twice in a row, and a new source behind a target
        c.set_target(tgt)
        c.set_source(ref[:, :3].copy())
        c.set_target(tgt)
        r2 = c.icp_align(max_corr=3.0, force_iterations=15, nn_method=wm.WM_NN_GRID, carry_state=0)
        out.append((r, r2, c.correspondences()))
        c.close()
    del os.environ["WM_TUNE_EARLY_SOURCE"]
    rd, _, cd = out[1]  # (the overlap off: the path every other host-cloud test has used since round 1)
    for r, r2, corr in out:
        assert r["rc"] == 0 and r2["rc"] == 0
        assert np.array_equal(r["T"], rd["T"]) and np.array_equal(r2["T"], rd["T"])
        assert r["n_corr"] == rd["n_corr"]
        assert np.array_equal(corr[0], cd[0]) and np.array_equal(corr[1], cd[1])
