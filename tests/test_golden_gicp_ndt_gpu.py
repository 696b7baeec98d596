"""The HIP path against the independent numpy / scipy restatement of PCL's NDT and GICP arithmetic
(tests/golden/gicp_ndt_golden.json): the same assertions the C oracle passes on the CPU
(tests/test_golden_gicp_ndt_cpu.py), made through the C ABI on the GPU."""
import numpy as np
import pytest

import golden_checks as G

pytestmark = pytest.mark.gpu
GOLD = G.golden()


@pytest.mark.parametrize("name", sorted(GOLD["ndt"]))
def test_hip_ndt_derivatives_match_the_independent_restatement(wm, ctx, testscan, name):
    c = GOLD["ndt"][name]
    target, _ = G.shifted(testscan, c["tx"])
    ctx.set_source(testscan[::c["source_stride"]])
    ctx.set_target(target)
    G.check_ndt_derivatives(c, lambda pose, sign: ctx.ndt_derivatives(pose, res=c["res"], pcl_d1_sign=sign))


@pytest.mark.parametrize("name", [n for n in sorted(GOLD["ndt"]) if "optimum_T" in GOLD["ndt"][n]])
def test_hip_ndt_reaches_the_independent_optimum(wm, ctx, testscan, name):
    c = GOLD["ndt"][name]
    target, _ = G.shifted(testscan, c["tx"])
    ctx.set_source(testscan)
    ctx.set_target(target)
    got = ctx.ndt_align(res=c["res"], step_size=3, max_iter=100, t_eps=1e-8)
    assert got["rc"] == 0 and got["converged"]
    dt, ang = G.pose_err(got["T"], c["optimum_T"])
    assert dt <= 1e-4 and ang <= 1e-4, (dt, ang)


@pytest.mark.parametrize("name", [n for n in sorted(GOLD["gicp"]) if "probe" in GOLD["gicp"][n]])
def test_hip_gicp_pieces_and_fixed_point(wm, ctx, testscan, name):
    c = GOLD["gicp"][name]
    target, P = G.shifted(testscan, c["tx"])
    if c["res"] > 0:  # pcl::VoxelGrid on the device (bit-exact vs the oracle: test_match_gpu.py)
        a, b = ctx.voxel_downsample(testscan, c["res"]), ctx.voxel_downsample(target, c["res"])
    else:
        a, b = testscan, target
    assert len(a) == c["n_ref"] and len(b) == c["n_target"]
    ctx.set_source(a)
    ctx.set_target(b)
    C1, C2 = ctx.gicp_covariances(k=10, eps=1e-3)
    np.testing.assert_allclose(C1[:4], c["cov_ref_first4"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(C2[:4], c["cov_target_first4"], rtol=0, atol=1e-9)
    assert abs(np.abs(C1).sum() - c["cov_ref_checksum"]) <= 1e-6 * c["cov_ref_checksum"]
    assert abs(np.abs(C2).sum() - c["cov_target_checksum"]) <= 1e-6 * c["cov_target_checksum"]
    f, g, m = ctx.gicp_eval(np.eye(4), np.array(c["probe"]["x"]))
    assert m == len(a)
    assert abs(f - c["probe"]["f"]) <= 2e-5 * abs(c["probe"]["f"])
    GG = np.array(c["probe"]["grad"])
    np.testing.assert_allclose(g, GG, rtol=1e-4, atol=2e-5 * np.abs(GG).max())
    got = ctx.gicp_align()
    assert got["rc"] == 0 and np.linalg.norm(got["T"] - P) < 0.1
    dt, ang = G.pose_err(got["T"], c["fixed_point_T"])
    assert dt <= 1e-4 and ang <= 1e-4, (dt, ang)


def test_hip_gicp_on_the_noisy_filtered_pair(wm, ctx, testscan):
    """As the oracle's test: wm_gicp_match with the voxel filter on the noisy pair, against the independent
    fixed point at the documented millimetre spread of PCL's early-stopping BFGS."""
    c = GOLD["gicp"]["noisyFiltered"]
    noisy, P = G.noisy_filtered_pair(testscan, c)
    got = ctx.gicp_match(testscan, noisy, res=c["res"])
    assert got["rc"] == 0 and got["n_corr"] == c["n_ref"] and np.linalg.norm(got["T"] - P) < 0.1
    dt, ang = G.pose_err(got["T"], c["fixed_point_T"])
    assert dt <= 3e-3 and ang <= 1e-3, (dt, ang)


def test_hip_ndt_pcl18_rule_takes_the_independent_newton_steps(wm, ctx, testscan):
    c = GOLD["ndt"]["smallDisplacement"]
    target, _ = G.shifted(testscan, c["tx"])
    ctx.set_source(testscan)
    ctx.set_target(target)
    steps = c["pcl18_newton_steps"]
    got = ctx.ndt_align(res=c["res"], step_size=3, max_iter=1, t_eps=1e-8, skip_line_search=1)
    assert got["rc"] == 0 and got["iterations"] == 3
    dt, ang = G.pose_err(got["T"], steps[2]["T"])
    assert dt <= 2e-4 and ang <= 2e-4, (dt, ang)
