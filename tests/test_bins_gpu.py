"""The iteration's sums as exact integer limbs in bins (libwave_amd/csrc/wm_bins.hpp, k_bins_solve): the unsharded ICP
loop's replacement for rows of partial sums + k_reduce_rows + k_reduce_solve.  The sums it solves from are the EXACT sum
of the waves' partial sums (the row path rounds after every addition, in a fixed order): registrations by the two paths
agree to the last few bits of the statistics, stop at the same iteration, and each is bit-reproducible -- the bins
without any fixed order of addition.  (pcl::IterativeClosestPoint::computeTransformation's per-iteration tail,
wave_matching/src/icp.cpp:95,116,126.)"""
import numpy as np
import pytest

from helpers import pose_error
from libwave_amd import synth

pytestmark = pytest.mark.gpu


def _align(wm, ref, tgt, bins, **kw):
    c = wm.Context(0)
    try:
        c.set_option("bins", bins)
        c.set_source(ref)
        c.set_target(tgt)
        kw.setdefault("nn_method", wm.WM_NN_GRID)
        return c.icp_align(max_corr=3.0, carry_state=0, **kw)
    finally:
        c.close()


@pytest.mark.parametrize("n", [60000, 200000])   # (938 rows: one solve kernel; 3 125 rows: k_reduce_rows first)
def test_bins_and_rows_give_the_same_registration(wm, n):
    ref, tgt, T_gt = synth.pair(n, seed=31, mode="resample")
    for kw in (dict(force_iterations=30), dict(max_iter=60)):      # forced (certified late iterations) and PCL's own stop
        a = _align(wm, ref, tgt, 1, **kw)
        b = _align(wm, ref, tgt, 0, **kw)
        assert a["rc"] == 0 and b["rc"] == 0
        assert a["iterations"] == b["iterations"] and a["n_corr"] == b["n_corr"]
        assert a["cert_launches"] == b["cert_launches"]
        dt, ang = pose_error(a["T"], b["T"])
        assert dt <= 1e-9 and ang <= 1e-10, (dt, ang)
        assert abs(a["mse"] - b["mse"]) <= 1e-12 * abs(b["mse"])


def test_bins_registration_is_bit_reproducible(wm):
    """No fixed order of addition anywhere across waves -- and the same bits every time: integer sums commute."""
    ref, tgt, _ = synth.pair(150000, seed=32, mode="resample")
    runs = [_align(wm, ref, tgt, 1, force_iterations=35) for _ in range(3)]
    assert all(r["rc"] == 0 for r in runs) and runs[0]["cert_launches"] > 0
    for r in runs[1:]:
        assert np.array_equal(r["T"], runs[0]["T"]) and r["mse"] == runs[0]["mse"]


def test_one_context_switching_between_bins_and_rows(wm):
    """The bins are all zero between registrations whatever ran before (the solve puts the zeros back)."""
    ref, tgt, _ = synth.pair(50000, seed=33, mode="resample")
    c = wm.Context(0)
    try:
        c.set_source(ref)
        c.set_target(tgt)
        outs = []
        for bins in (1, 0, 1, 1, 0):
            c.set_option("bins", bins)
            outs.append(c.icp_align(max_corr=3.0, force_iterations=12, nn_method=wm.WM_NN_GRID, carry_state=0))
        assert all(o["rc"] == 0 for o in outs)
        assert np.array_equal(outs[0]["T"], outs[2]["T"]) and np.array_equal(outs[2]["T"], outs[3]["T"])
        assert np.array_equal(outs[1]["T"], outs[4]["T"])
    finally:
        c.close()


def test_sums_the_limbs_cannot_hold_end_the_registration_loudly(wm):
    """Coordinates of 3e9 m: a wave's sum of products is ~6e20, beyond the limbs' 2^62 -- the bin is poisoned and the
    registration ends with "not enough correspondences" instead of solving from a wrapped-around integer."""
    rng = np.random.default_rng(4)
    ref = (rng.uniform(-40, 40, (20000, 3)) + 3.0e9).astype(np.float32)
    got = _align(wm, ref, ref.copy(), 1, force_iterations=4)
    assert got["rc"] == wm.WM_TOO_FEW
