"""GICP's objective as sufficient statistics (libwave_amd/csrc/wm_gicp_quad.hpp; wm_gicp_params::objective =
WM_GICP_OBJECTIVE_STATISTICS, an explicit opt-in) against PCL's per-pair sums (WM_GICP_OBJECTIVE_PCL_SUMS, the default -- what
OptimizationFunctorWithIndices::fdf does; the reference reaches it through wave_matching/src/gicp.cpp:58).

Bit-level parity of EACH objective with the oracle's restatement of it: tests/test_gicp_gpu.py.  Here: how far the two
objectives' registrations are apart.  They are the same function of the transform up to the float rounding PCL's
per-point transform adds away from the pairing transform (a relative ~4e-7 of f); PCL's BFGS stops at a gradient
tolerance of 1e-2 wherever its line search lands, so that dust moves the stopping point exactly as PCL's own summation
order does (test_gicp_gpu.py::test_gicp_spread_against_pcl_literal_summation): nothing on pairs that register sharply,
1e-5 .. 1e-3 m on noisy re-sampled pairs, neither objective closer to ground truth."""
import numpy as np
import pytest

from helpers import pose_error
from libwave_amd import synth

pytestmark = pytest.mark.gpu

CASES = [("fullResNullMatch", -1.0, 0.0), ("nullDisplacement", 0.05, 0.0), ("smallDisplacement", 0.05, 0.2),
         ("smallDisplacementFullRes", -1.0, 0.2)]


@pytest.mark.parametrize("name,res,tx", CASES)
def test_reference_cases_both_objectives(wm, ctx, oracle, testscan, name, res, tx):
    """wave_matching/tests/gicp_tests.cpp:43-100 (threshold :36): both objectives meet the reference's own bar, and
    each other within 5e-4 m / 1e-4 rad (the voxel-filtered 0.2 m case is where they differ most: the per-pair
    objective's line search gives up on its own float dust 2e-4 m short of the exact answer, the statistics carry on
    to it)."""
    P = np.eye(4)
    P[0, 3] = tx
    target = oracle.transform_cloud_d(testscan, P)
    a = ctx.gicp_match(testscan, target, res=res, objective=wm.WM_GICP_OBJECTIVE_STATISTICS)
    b = ctx.gicp_match(testscan, target, res=res, objective=wm.WM_GICP_OBJECTIVE_PCL_SUMS)
    assert a["rc"] == 0 and b["rc"] == 0
    assert np.linalg.norm(a["T"] - P) < 0.1 and np.linalg.norm(b["T"] - P) < 0.1
    dt, ang = pose_error(a["T"], b["T"])
    ea, eb = pose_error(a["T"], P)[0], pose_error(b["T"], P)[0]
    print("GICP %s: statistics vs per-pair sums %.2e m / %.2e rad | vs ground truth %.2e / %.2e m | outer iterations %d / %d, "
          "evaluations %d / %d" % (name, dt, ang, ea, eb, a["iterations"], b["iterations"], a["evaluations"], b["evaluations"]))
    assert dt <= 5e-4 and ang <= 1e-4, (dt, ang)
    assert ea <= eb + 1e-4          # (never farther from the exact answer of a copy pair than the per-pair objective)


def test_spread_between_the_objectives_on_noisy_pairs(wm, ctx):
    rows = []
    for n, seed in ((30000, 21), (20000, 7), (40000, 5), (20000, 101), (5000, 103)):
        ref, tgt, T_gt = synth.pair(n, seed=seed)
        ctx.set_source(ref)
        ctx.set_target(tgt)
        a = ctx.gicp_align(objective=wm.WM_GICP_OBJECTIVE_STATISTICS)
        b = ctx.gicp_align(objective=wm.WM_GICP_OBJECTIVE_PCL_SUMS)
        assert a["rc"] == 0 and b["rc"] == 0 and a["n_corr"] > 0.9 * n
        dt, ang = pose_error(a["T"], b["T"])
        ea, eb = pose_error(a["T"], T_gt)[0], pose_error(b["T"], T_gt)[0]
        rows.append((n, seed, dt, ang, ea, eb, a["iterations"], b["iterations"], a["evaluations"], b["evaluations"]))
        assert dt < 5e-3 and ang < 1e-3
        assert abs(ea - eb) < 5e-3
    for r in rows:
        print("GICP %d pts seed %d: statistics vs per-pair sums %.2e m / %.2e rad | vs ground truth %.2e / %.2e m | "
              "outer iterations %d / %d, evaluations %d / %d" % r)
    # neither objective is systematically closer to the truth
    assert abs(np.mean([r[4] for r in rows]) - np.mean([r[5] for r in rows])) < 1e-3


def test_statistics_objective_needs_no_pass_per_evaluation(wm, ctx):
    """What the form buys: the device time spent on the objective is one pass per OUTER iteration (k_gicp_quad),
    whatever the number of evaluations the line searches ask for."""
    import os
    ref, tgt, _ = synth.pair(60000, seed=11)
    os.environ["WM_GICP_PROFILE"] = "1"
    try:
        prof = wm.Context(0)
    finally:
        del os.environ["WM_GICP_PROFILE"]
    try:
        prof.set_source(ref)
        prof.set_target(tgt)
        a = prof.gicp_align(objective=wm.WM_GICP_OBJECTIVE_STATISTICS)
        b = prof.gicp_align(objective=wm.WM_GICP_OBJECTIVE_PCL_SUMS)
    finally:
        prof.close()
    assert a["rc"] == 0 and b["rc"] == 0
    assert a["served_evaluations"] == 0
    print("objective kernels: statistics %.3f ms for %d evaluations (%d outer iterations); per-pair sums %.3f ms for %d"
          % (a["fdf_kernel_ms"], a["evaluations"], a["iterations"], b["fdf_kernel_ms"], b["evaluations"]))
    assert a["fdf_kernel_ms"] < 0.5 * b["fdf_kernel_ms"]


def test_both_objectives_against_the_independent_fixed_point(wm, ctx, testscan):
    """tests/golden/gicp_ndt_golden.json (make_golden_gicp_ndt.py: numpy / scipy, written without either path in
    view) holds the FIXED POINT of pair -> minimise (scipy BFGS to 1e-12, double-precision transform) -> re-pair for the
    reference's cases and a noisy voxel-filtered pair: where a GICP that converged fully would land.  Both objectives
    are held to it at the golden tests' bars; the statistics objective -- the same smooth function scipy minimised -- is
    never farther from it than PCL's per-pair objective is (plus the float quantum of the result)."""
    import golden_checks as G
    GOLD = G.golden()["gicp"]
    for name in ("smallDisplacement", "fullResSmallDisplacement", "noisyFiltered"):
        c = GOLD[name]
        if name == "noisyFiltered":
            target, P = G.noisy_filtered_pair(testscan, c)
            bar = (3e-3, 1e-3)
        else:
            target, P = G.shifted(testscan, c["tx"])
            bar = (1e-4, 1e-4)
        a = ctx.gicp_match(testscan, target, res=c["res"], objective=wm.WM_GICP_OBJECTIVE_STATISTICS)
        b = ctx.gicp_match(testscan, target, res=c["res"], objective=wm.WM_GICP_OBJECTIVE_PCL_SUMS)
        assert a["rc"] == 0 and b["rc"] == 0
        da, ra = pose_error(a["T"], np.array(c["fixed_point_T"]))
        db, rb = pose_error(b["T"], np.array(c["fixed_point_T"]))
        print("GICP %s vs the independent fixed point: statistics %.2e m / %.2e rad, per-pair sums %.2e m / %.2e rad"
              % (name, da, ra, db, rb))
        assert da <= bar[0] and ra <= bar[1] and db <= bar[0] and rb <= bar[1]
        assert da <= db + 2e-5


def test_one_context_alternating_between_the_objectives(wm):
    """A context that ran PCL's per-pair objective first (its resident evaluator allocates the pinned answer buffer)
    and then the statistics objective (which fetches 74 sums into the same buffer), and back: each result is the one
    a fresh context gives for that objective."""
    ref, tgt, _ = synth.pair(20000, seed=5, mode="resample")
    fresh = {}
    for obj in (wm.WM_GICP_OBJECTIVE_PCL_SUMS, wm.WM_GICP_OBJECTIVE_STATISTICS):
        c = wm.Context(0)
        c.set_source(ref)
        c.set_target(tgt)
        fresh[obj] = c.gicp_align(objective=obj)
        c.close()
    c = wm.Context(0)
    c.set_source(ref)
    c.set_target(tgt)
    for obj in (wm.WM_GICP_OBJECTIVE_PCL_SUMS, wm.WM_GICP_OBJECTIVE_STATISTICS, wm.WM_GICP_OBJECTIVE_PCL_SUMS,
                wm.WM_GICP_OBJECTIVE_STATISTICS):
        got = c.gicp_align(objective=obj)
        assert got["rc"] == fresh[obj]["rc"] == 0
        assert np.array_equal(got["T"], fresh[obj]["T"]) and got["iterations"] == fresh[obj]["iterations"]
    c.close()


def test_the_default_objective_is_pcls(wm, ctx, oracle, testscan):
    """The drop-in default is the REFERENCE's algorithm (wave_matching/src/gicp.cpp:58 -> PCL's per-pair objective): a
    registration with default parameters equals, bit for bit, one that asks for WM_GICP_OBJECTIVE_PCL_SUMS, and the
    oracle's default mode; the statistics objective is only ever run when asked for."""
    assert wm.WM_GICP_OBJECTIVE_PCL_SUMS == 0 and wm.gicp_params().objective == wm.WM_GICP_OBJECTIVE_PCL_SUMS
    P = np.eye(4)
    P[0, 3] = 0.2
    target = oracle.transform_cloud_d(testscan, P)
    a = ctx.gicp_match(testscan, target, res=0.05)
    b = ctx.gicp_match(testscan, target, res=0.05, objective=wm.WM_GICP_OBJECTIVE_PCL_SUMS)
    oracle.gicp_set_objective(0)
    want = oracle.gicp_align(oracle.voxel_grid(testscan, 0.05), oracle.voxel_grid(target, 0.05))
    assert a["rc"] == b["rc"] == 0
    assert np.array_equal(a["T"], b["T"]) and a["evaluations"] == b["evaluations"] and a["served_evaluations"] > 0
    assert np.array_equal(a["T"], want["T"]) and a["iterations"] == want["iterations"]
    with pytest.raises(wm.WmError):
        ctx.gicp_align(params=wm.gicp_params(objective=7))
