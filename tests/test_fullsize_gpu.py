"""BASELINE.json's full-size configurations on one MI355X, checked through properties that do not
need the (far too slow) CPU oracle at that size:

  configs[1]  ICP 1M<->1M: the grid search equals the all-pairs search on a query sample against
              the FULL 1M target (bit-exact); 50 forced iterations land on the ground truth; a
              warm search at the final pose equals a cold one; an exact copy registers to 1e-6.
  configs[2]  GICP 500k<->500k recovers the ground-truth transform.
  configs[3]  NDT 2M<->2M at 0.5 m voxels recovers the ground-truth transform -- on the area-uniform
              scene and on its 64-ring lidar sampling (KITTI-like: what bench.py times); ICP on that
              sampling: grid = all-pairs on a sample, certified correspondences included.
  configs[4]  ICP 8M<->8M in 8 target slabs: one registration's statistics block, accumulated
              slab by slab on one GPU, equals the unsharded block (what the all-reduce would
              deliver), and every source point is owned by exactly one slab.
"""
import numpy as np
import pytest
import torch

from helpers import pose_error
from libwave_amd import sharding, synth

pytestmark = pytest.mark.gpu


def test_icp_1m_grid_equals_all_pairs_on_a_sample(wm, ctx):
    ref, tgt, T_gt = synth.pair(1_000_000, seed=42)
    sample = np.ascontiguousarray(ref[np.random.default_rng(0).choice(len(ref), 30000, replace=False)])
    ctx.set_source(sample)
    ctx.set_target(torch.from_numpy(tgt).cuda())
    for T in (np.eye(4), T_gt):
        gi, gd = ctx.nn_search(T, 3.0, wm.WM_NN_GRID)
        bi, bd = ctx.nn_search(T, 3.0, wm.WM_NN_BRUTE)
        assert np.array_equal(gi, bi)
        assert np.array_equal(gd[gi >= 0], bd[bi >= 0])
        assert (gi >= 0).mean() > 0.99


def test_icp_1m_registration_properties(wm, ctx):
    ref, tgt, T_gt = synth.pair(1_000_000, seed=42)
    d_ref, d_tgt = torch.from_numpy(ref).cuda(), torch.from_numpy(tgt).cuda()
    ctx.set_source(d_ref)
    ctx.set_target(d_tgt)
    r = ctx.icp_align(max_corr=3.0, force_iterations=50, nn_method=wm.WM_NN_GRID, carry_state=0)
    assert r["rc"] == 0 and r["iterations"] == 50 and r["n_corr"] > 990_000
    dt, ang = pose_error(r["T"], T_gt)
    assert dt < 5e-4 and ang < 5e-5, (dt, ang)   # resampled + 1 cm noise: the MSE floor, not 0
    # the same registration twice is the same registration (fixed-order reductions)
    ctx.set_source(d_ref)
    ctx.set_target(d_tgt)
    r2 = ctx.icp_align(max_corr=3.0, force_iterations=50, nn_method=wm.WM_NN_GRID, carry_state=0)
    assert np.array_equal(r["T"], r2["T"]) and r["mse"] == r2["mse"]
    # warm (seeded) and cold searches at the final pose agree bit for bit
    ci, cd = ctx.nn_search(r["T"], 3.0, wm.WM_NN_GRID)
    wi, wd = ctx.nn_search(r["T"], 3.0, wm.WM_NN_GRID | wm.WM_NN_WARM)
    assert np.array_equal(ci, wi) and np.array_equal(cd, wd)
    # statistics of those matches: n and sum d^2 consistent with the reported MSE
    st = ctx.icp_stats_for(r["T"])
    assert st[0] == (ci >= 0).sum()
    assert abs(st[16] - cd[ci >= 0].astype(np.float64).sum()) <= 1e-9 * st[16]


def test_icp_1m_exact_copy(wm, ctx):
    ref, tgt, T_gt = synth.pair(1_000_000, seed=7, mode="copy")
    ctx.set_source(torch.from_numpy(ref).cuda())
    ctx.set_target(torch.from_numpy(tgt).cuda())
    r = ctx.icp_align(max_corr=3.0, max_iter=100, t_eps=1e-12, fit_eps=1e-12, nn_method=wm.WM_NN_GRID,
                      carry_state=0)
    assert r["rc"] == 0
    dt, ang = pose_error(r["T"], T_gt)
    assert dt < 1e-5 and ang < 1e-6, (dt, ang, r["iterations"])


def test_gicp_500k_recovers_ground_truth(wm, ctx):
    ref, tgt, T_gt = synth.pair(500_000, seed=42)
    ctx.set_source(torch.from_numpy(ref).cuda())
    ctx.set_target(torch.from_numpy(tgt).cuda())
    r = ctx.gicp_align()
    assert r["rc"] == 0 and r["converged"]
    dt, ang = pose_error(r["T"], T_gt)
    # 3e-3 m: on a resampled, noisy pair PCL's BFGS stops wherever its line search lands in the
    # sliding directions (see test_gicp_on_noisy_synthetic_pair); the rotation is well constrained
    assert dt < 3e-3 and ang < 1e-4, (dt, ang)


def test_ndt_2m_recovers_ground_truth(wm, ctx):
    ref, tgt, T_gt = synth.pair(2_000_000, seed=42)
    ctx.set_source(torch.from_numpy(ref).cuda())
    ctx.set_target(torch.from_numpy(tgt).cuda())
    r = ctx.ndt_align(res=0.5)
    assert r["rc"] == 0 and r["converged"] and r["n_voxels"] > 10000
    dt, ang = pose_error(r["T"], T_gt)
    assert dt < 1e-3 and ang < 1e-4, (dt, ang)


def test_ndt_2m_64_ring_scan_recovers_ground_truth(wm, ctx):
    """BASELINE configs[3] says KITTI-like: the 64-ring lidar sampling of the scene bench.py uses (dense
    next to the sensor -- 1 700 points in the voxels there -- sparse far out), not the area-uniform one."""
    ref, tgt, T_gt = synth.pair(2_000_000, seed=42, pattern="rings")
    ctx.set_source(torch.from_numpy(ref).cuda())
    ctx.set_target(torch.from_numpy(tgt).cuda())
    r = ctx.ndt_align(res=0.5)
    assert r["rc"] == 0 and r["converged"] and r["n_voxels"] > 10000
    dt, ang = pose_error(r["T"], T_gt)
    assert dt < 1e-3 and ang < 1e-4, (dt, ang)


def test_icp_1m_64_ring_scan_grid_equals_all_pairs_on_a_sample(wm, ctx):
    """Non-uniform density (hundreds of points of one scan line in a cell near the sensor, empty cells far
    out): the grid search -- full and certified -- against the all-pairs search on a sample of the
    queries, bit for bit; and the 50-iteration registration lands on ground truth."""
    ref, tgt, T_gt = synth.pair(1_000_000, seed=42, pattern="rings")
    d_tgt = torch.from_numpy(tgt).cuda()
    sel = np.random.default_rng(5).choice(len(ref), 30000, replace=False)
    ctx.set_source(ref[sel])
    ctx.set_target(d_tgt)
    gi, gd = ctx.nn_search(np.eye(4), max_corr=3.0, nn_method=wm.WM_NN_GRID)
    bi, bd = ctx.nn_search(np.eye(4), max_corr=3.0, nn_method=wm.WM_NN_BRUTE)
    assert np.array_equal(gi, bi) and np.array_equal(gd, bd)
    ctx.set_source(torch.from_numpy(ref).cuda())
    ctx.set_target(d_tgt)
    r = ctx.icp_align(max_corr=3.0, force_iterations=50, nn_method=wm.WM_NN_GRID, carry_state=0)
    assert r["rc"] == 0 and r["cert_launches"] > 0
    dt, ang = pose_error(r["T"], T_gt)
    assert dt < 1e-3 and ang < 1e-4, (dt, ang)
    # the correspondences the certified iterations left = an all-pairs search of a sample under the final pose's predecessor
    ci, cd = ctx.correspondences()
    r49 = ctx.icp_align(max_corr=3.0, force_iterations=49, nn_method=wm.WM_NN_GRID, carry_state=0)
    ctx.set_source(ref[sel])
    bi, bd = ctx.nn_search(r49["T"], max_corr=3.0, nn_method=wm.WM_NN_BRUTE)
    assert np.array_equal(ci[sel], bi)


def test_icp_8m_slab_statistics_add_up(wm, ctx):
    world = 8
    ref, tgt, T_gt = synth.pair_tiled(1_000_000, world, seed=42)
    I = np.eye(4)
    ctx.set_source(torch.from_numpy(ref).cuda())
    ctx.set_target(torch.from_numpy(tgt).cuda())
    ctx.nn_search(I, 3.0, wm.WM_NN_GRID, want=False)
    whole = ctx.icp_stats_for(I)                       # the registration's first statistics block
    assert whole[31] == len(ref)
    total = np.zeros_like(whole)
    p = wm.icp_params(max_corr=3.0, force_iterations=1, nn_method=wm.WM_NN_GRID)
    for rank in range(world):
        eng = sharding.GpuShardEngine(0, ref, tgt, rank, world, 3.0)
        assert eng.n_target_local < 0.2 * len(tgt) and eng.n_source_local < 0.3 * len(ref)
        eng.begin(p)                                   # starts at identity, like wm_icp_align
        total += eng.local_stats().cpu().numpy()
        del eng
    assert total[31] == len(ref)                       # every source point owned exactly once
    assert total[0] == whole[0]                        # same number of correspondences
    np.testing.assert_allclose(total[:17], whole[:17], rtol=1e-9, atol=1e-6)
