"""The certificate kernel (k_nn_cert, csrc/wm_nn.hip): late ICP iterations prove per query that the
previous match is still the nearest neighbour and search only where the proof fails.  It must give
the SAME correspondences as a search of every query -- checked against the kd-tree oracle bit for
bit -- and the same registration as the all-search path.  Replaces
pcl::registration::CorrespondenceEstimation::determineCorrespondences (wave_matching/src/icp.cpp:126)."""
import numpy as np
import pytest

from helpers import pose_error
from libwave_amd import synth

pytestmark = pytest.mark.gpu


def _align(wm, ref, tgt, iters, cert_from, max_corr=3.0, **opts):
    c = wm.Context(0)
    try:
        c.set_option("cert_from", cert_from)
        for k, v in opts.items():
            c.set_option(k, v)
        c.set_source(ref)
        c.set_target(tgt)
        r = c.icp_align(max_corr=max_corr, force_iterations=iters, nn_method=wm.WM_NN_GRID, carry_state=0)
        r["corr"] = c.correspondences()
        return r
    finally:
        c.close()


def _oracle_corr(oracle, ref, tgt, T, max_corr):
    moved = oracle.transform_cloud_f(ref, T.astype(np.float32))
    oi, od = oracle.KdTree(tgt).nn(moved)
    keep = od.astype(np.float64) <= float(max_corr) ** 2
    return np.where(keep, oi, -1), od, keep


def _cases():
    out = []
    ref, tgt, _ = synth.pair(30000, seed=5, mode="resample")
    out.append(("resample30k", ref, tgt, 3.0))
    ref, tgt, _ = synth.pair(20000, seed=6, mode="copy")
    out.append(("copy20k", ref, tgt, 3.0))
    # partial overlap: a third of the source has nothing within max_corr
    ref, tgt, _ = synth.pair(24000, seed=8, mode="resample")
    out.append(("partial", ref, tgt[tgt[:, 0] < 15.0], 1.0))
    # lattice target: many exact ties in distance (lowest index must win, certified or searched)
    g = np.stack(np.meshgrid(np.arange(40), np.arange(40), np.arange(4), indexing="ij"), -1).reshape(-1, 3)
    lat = (g * 0.25).astype(np.float32)
    rng = np.random.Generator(np.random.PCG64(3))
    q = (lat[rng.permutation(len(lat))[:5000]] + np.float32(0.125)).astype(np.float32)
    out.append(("lattice-ties", q, lat, 1.0))
    # UTM-like coordinates: the pair of the first case 1e4 .. 5e4 m from the origin -- 4 mm float
    # spacing in y, where the grid search had its stress case (test_nn_stress_gpu.py) and the certificate's cushions
    # (1e-4 relative + 1e-6 m on either side of the comparison) had only been argued
    ref, tgt, T = synth.pair(30000, seed=5, mode="resample")
    off = np.array([12345.0, -54321.0, 250.0])
    ref_o = (ref.astype(np.float64) + off).astype(np.float32)
    # (the same rigid motion about the shifted scene's own centre: x -> R (x - off) + t + off)
    other = (tgt.astype(np.float64) - T[:3, 3]) @ T[:3, :3]          # undo T: the re-sampled, noisy scene
    tgt_o = ((other @ T[:3, :3].T) + T[:3, 3] + off).astype(np.float32)
    out.append(("utm-offset", ref_o, tgt_o, 3.0))
    return out


@pytest.mark.parametrize("name,ref,tgt,max_corr", _cases(), ids=[c[0] for c in _cases()])
@pytest.mark.parametrize("cert_from", [0, 1, 4, -1])
def test_certified_correspondences_are_the_kdtree_s(wm, oracle, name, ref, tgt, max_corr, cert_from):
    """Correspondences left by a K-iteration align belong to the pose after K - 1 iterations (the
    same configuration run K - 1 iterations gives it: the path is bit-reproducible): they must be
    the exact nearest neighbours under that pose, certified or searched."""
    K = 14
    a = _align(wm, ref, tgt, K - 1, cert_from, max_corr)
    b = _align(wm, ref, tgt, K, cert_from, max_corr)
    assert a["rc"] == 0 and b["rc"] == 0
    if cert_from >= 0:
        assert b["cert_launches"] == K - cert_from
    want_i, want_d, keep = _oracle_corr(oracle, ref, tgt, a["T"], max_corr)
    gi, gd = b["corr"]
    assert np.array_equal(gi, want_i), "%d of %d matches differ" % ((gi != want_i).sum(), len(gi))
    assert np.array_equal(gd[keep], want_d[keep])


@pytest.mark.parametrize("cert_from", [0, 3, -1])
def test_certified_registration_equals_the_all_search_one(wm, cert_from):
    ref, tgt, _ = synth.pair(40000, seed=12, mode="resample")
    full = _align(wm, ref, tgt, 30, -2)
    cert = _align(wm, ref, tgt, 30, cert_from)
    assert full["cert_launches"] == 0
    assert cert["cert_launches"] > 0
    dt, da = pose_error(full["T"], cert["T"])
    assert dt < 1e-9 and da < 1e-10, (dt, da)   # the sums differ in order only
    assert cert["n_corr"] == full["n_corr"]
    assert np.array_equal(full["corr"][0], cert["corr"][0])


def test_most_queries_settle_once_aligned(wm):
    """The point of the exercise: in the aligned state nearly every query is settled by its bound."""
    ref, tgt, _ = synth.pair(60000, seed=14, mode="resample")
    c = wm.Context(0)
    try:
        c.set_option("cert_from", 0)
        c.cert_log(64)
        c.set_source(ref)
        c.set_target(tgt)
        r = c.icp_align(max_corr=3.0, force_iterations=40, nn_method=wm.WM_NN_GRID, carry_state=0)
        log = c.cert_log()
    finally:
        c.close()
    assert r["rc"] == 0 and len(log) == 40
    assert log[0] == 60000 and log[1] == 60000      # no bounds yet: everything is searched
    assert log[-1] < 0.15 * 60000, log


def test_certificate_survives_stopping_rules_and_gn6(wm, oracle):
    """Free-running (PCL's stopping rules) and the Gauss-Newton step: same stop, same result."""
    ref, tgt, _ = synth.pair(30000, seed=15, mode="resample")
    for mode in (wm.WM_ICP_SVD, wm.WM_ICP_GN6):
        res = []
        for cert_from in (-2, 0, -1):
            c = wm.Context(0)
            try:
                c.set_option("cert_from", cert_from)
                c.set_source(ref)
                c.set_target(tgt)
                res.append(c.icp_align(max_corr=3.0, max_iter=60, t_eps=1e-12, fit_eps=1e-9, mode=mode,
                                       nn_method=wm.WM_NN_GRID, carry_state=0))
            finally:
                c.close()
        for r in res[1:]:
            assert r["iterations"] == res[0]["iterations"] and r["state"] == res[0]["state"]
            dt, da = pose_error(r["T"], res[0]["T"])
            assert dt < 1e-9 and da < 1e-10


def test_certified_align_is_bit_reproducible(wm):
    ref, tgt, _ = synth.pair(50000, seed=16, mode="resample")
    a = _align(wm, ref, tgt, 35, -1)
    b = _align(wm, ref, tgt, 35, -1)
    assert a["cert_launches"] == b["cert_launches"] > 0
    assert np.array_equal(a["T"], b["T"])
    assert np.array_equal(a["corr"][1], b["corr"][1])


# ---- the resident form (k_nn_cert<.., LATE> + k_late_solver, option "late"): the late iterations, their sums, the
# solve and the stopping rules in ONE launch.  Off by default (it is not faster: profiles/r04_experiments.md);
# it must give the launched path's correspondences and registration all the same.
@pytest.mark.parametrize("name,ref,tgt,max_corr", _cases(), ids=[c[0] for c in _cases()])
@pytest.mark.parametrize("cert_from", [1, 4, -1])
def test_resident_kernel_leaves_the_kdtree_s_correspondences(wm, oracle, name, ref, tgt, max_corr, cert_from):
    K = 14
    a = _align(wm, ref, tgt, K - 1, cert_from, max_corr, late=1)
    b = _align(wm, ref, tgt, K, cert_from, max_corr, late=1)
    assert a["rc"] == 0 and b["rc"] == 0
    if cert_from >= 0:
        assert b["cert_launches"] == K - cert_from and b["late_iterations"] == K - cert_from
    gi, gd = b["corr"]
    want_i, want_d, keep = _oracle_corr(oracle, ref, tgt, a["T"], max_corr)
    assert np.array_equal(gi, want_i), "%d of %d matches differ" % ((gi != want_i).sum(), len(gi))
    assert np.array_equal(gd[keep], want_d[keep])


def test_resident_kernel_registers_like_the_launched_path(wm, oracle):
    ref, tgt, _ = synth.pair(60000, seed=11, mode="resample")
    for kw in ({"force_iterations": 40}, {"max_iter": 60, "t_eps": 1e-12, "fit_eps": 1e-9}):
        res = []
        for late in (0, 1):
            c = wm.Context(0)
            c.set_option("late", late)
            c.set_source(ref)
            c.set_target(tgt)
            r = c.icp_align(max_corr=3.0, nn_method=wm.WM_NN_GRID, carry_state=0, **kw)
            r["corr"] = c.correspondences()
            res.append(r)
            c.close()
        a, b = res
        assert a["rc"] == 0 and b["rc"] == 0
        assert b["late_iterations"] > 0 and a["late_iterations"] == 0
        assert a["iterations"] == b["iterations"] and a["state"] == b["state"]
        dt, da = pose_error(a["T"], b["T"])
        assert dt < 1e-9 and da < 1e-10, (dt, da)   # the sums differ in order only
        assert np.array_equal(a["corr"][0], b["corr"][0])
    # bit-reproducible
    x = _align(wm, ref, tgt, 30, -1, late=1)
    y = _align(wm, ref, tgt, 30, -1, late=1)
    assert x["late_iterations"] == y["late_iterations"] > 0
    assert np.array_equal(x["T"], y["T"]) and np.array_equal(x["corr"][1], y["corr"][1])
