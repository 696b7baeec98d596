"""GPU parity: VoxelGrid, transformPointCloud and the whole ICPMatcher::match()
(wave_matching/src/icp.cpp:75-133, all three branches) through the C ABI vs the oracle,
on the reference's own fixture and test perturbations (tests/icp_tests.cpp)."""
import json
import os

import numpy as np
import pytest

from helpers import TOL_R, TOL_T, pose_error
from libwave_amd import synth

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


# fine leaves take the lane-per-leaf centroid kernel, coarse ones (> 24 points per leaf on
# average) the wave-per-leaf one; 5 m and 40 m leaves hold thousands of points each
@pytest.mark.parametrize("leaf", [0.05, 0.1, 0.4, 0.8, 5.0, 40.0])
def test_voxel_grid_bit_exact(ctx, oracle, testscan, leaf):
    got = ctx.voxel_downsample(testscan, leaf)
    want = oracle.voxel_grid(testscan, leaf)
    assert got.shape == want.shape
    assert np.array_equal(got, want)   # float centroids, same order: bit-identical


def test_voxel_grid_synthetic_and_edge_cases(ctx, oracle):
    pts = synth.scene(200000, seed=3)
    assert np.array_equal(ctx.voxel_downsample(pts, 0.25), oracle.voxel_grid(pts, 0.25))
    assert np.array_equal(ctx.voxel_downsample(pts, 3.0), oracle.voxel_grid(pts, 3.0))   # big leaves
    bad = pts[:5000].copy()
    bad[::50, 0] = np.nan
    assert np.array_equal(ctx.voxel_downsample(bad, 0.5), oracle.voxel_grid(bad, 0.5))
    assert len(ctx.voxel_downsample(pts[:0], 0.1)) == 0
    one = ctx.voxel_downsample(pts[:1], 0.1)
    assert np.array_equal(one, pts[:1])
    # index space overflows int32 -> PCL returns the input unfiltered
    huge = np.array([[0, 0, 0], [3000, 3000, 3000]], np.float32)
    assert np.array_equal(ctx.voxel_downsample(huge, 0.001), huge)


def test_transform_cloud_bit_exact(ctx, oracle, testscan):
    T = synth.make_T((0.2, -3.0, 1.5), (0.3, -0.2, 1.1))
    assert np.array_equal(ctx.transform_cloud(testscan, T), oracle.transform_cloud_d(testscan, T))


CASES = {  # name: (res, multiscale_steps, tx) -- wave_matching/tests/icp_tests.cpp
    "fullResNullMatch": (-1.0, 0, 0.0),
    "nullDisplacement": (0.05, 0, 0.0),
    "smallDisplacement": (0.05, 0, 0.2),
    "multiscale": (0.1, 3, 0.2),
}


@pytest.mark.parametrize("case", sorted(CASES))
def test_reference_icp_match_cases(wm, ctx, oracle, testscan, case):
    res, steps, tx = CASES[case]
    with open(os.path.join(HERE, "golden", "icp_golden.json")) as f:
        gold = json.load(f)["cases"][case]
    perturb = np.eye(4)
    perturb[0, 3] = tx
    target = oracle.transform_cloud_d(testscan, perturb)
    got = ctx.icp_match(testscan, target, res=res, multiscale_steps=steps)
    assert got["rc"] == 0 and got["converged"]
    # the reference test's assertion (threshold, icp_tests.cpp:37)
    assert np.linalg.norm(got["T"] - perturb) < 0.1
    # parity with the oracle / the independent golden
    want = oracle.IcpMatch(testscan, target, res=res, multiscale_steps=steps, incremental_float=0)
    assert want.ok
    assert got["iterations"] == want.r.iterations
    dt, ang = pose_error(got["T"], want.T)
    assert dt <= 1e-6 and ang <= 1e-7, (dt, ang)
    dt, ang = pose_error(got["T"], np.array(gold["T"]))
    assert dt <= 1e-6 and ang <= 1e-7
    lit = oracle.IcpMatch(testscan, target, res=res, multiscale_steps=steps, incremental_float=1,
                          float_sums=1)
    dt, ang = pose_error(got["T"], lit.T)
    assert dt <= TOL_T and ang <= TOL_R


def test_match_on_synthetic_multiscale(wm, ctx, oracle):
    ref, tgt, T_gt = synth.pair(60000, seed=5, mode="resample")
    got = ctx.icp_match(ref, tgt, res=0.1, multiscale_steps=2, max_corr=3.0)
    want = oracle.IcpMatch(ref, tgt, res=0.1, multiscale_steps=2, incremental_float=0)
    assert got["rc"] == 0 and want.ok
    dt, ang = pose_error(got["T"], want.T)
    assert dt <= 1e-6 and ang <= 1e-7, (dt, ang)


def test_match_failure_leaves_result_untouched(wm, ctx):
    a = synth.scene(2000, seed=1)
    got = ctx.icp_match(a, a + np.float32(100.0), res=0.5, multiscale_steps=1)
    assert got["rc"] == wm.WM_TOO_FEW and got["T"] is None


@pytest.mark.parametrize("case", ["smallDisplacement", "multiscale"])
def test_information_matrices_after_match_against_golden(wm, ctx, oracle, testscan, case):
    """match() + estimateLUM / estimateLUMold on the device against the independent numpy vectors
    (tests/golden/icp_golden.json: info_lum / info_lumold, made by make_golden.py's lum_info) for the
    voxel-filtered and the multiscale branch.  The residual s^2 of these exact-copy cases is pure
    rounding noise (1e-9): the reference sums it as a sequential float, the kernel adds the same float
    terms in double -- hence 1e-3 on the matrix, exact on its structure (M'M itself to 1e-9)."""
    res, steps, tx = CASES[case]
    with open(os.path.join(HERE, "golden", "icp_golden.json")) as f:
        gold = json.load(f)["cases"][case]
    perturb = np.eye(4)
    perturb[0, 3] = tx
    target = oracle.transform_cloud_d(testscan, perturb)
    got = ctx.icp_match(testscan, target, res=res, multiscale_steps=steps)
    assert got["rc"] == 0
    for method, key in ((wm.WM_INFO_LUM, "info_lum"), (wm.WM_INFO_LUMOLD, "info_lumold")):
        rc, info, deg = ctx.icp_info(method, got["T"], max_corr=3.0)
        want = np.array(gold[key]["M"])
        assert rc == 0 and not deg and info[0, 0] > 0
        np.testing.assert_allclose(info, want, rtol=1e-3, atol=1e-9 * np.abs(want).max())
        # s^2 cancels in the ratio of two entries: the normal equations themselves agree to 1e-9
        np.testing.assert_allclose(info / info[0, 0], want / want[0, 0], rtol=1e-7, atol=1e-9)
