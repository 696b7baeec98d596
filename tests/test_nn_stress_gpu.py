"""Correspondence search stress cases (GPU, through the C ABI) against the oracle's exact
kd-tree: every index and every squared distance must be bit-identical, for cold searches and
for warm ones (seeded with the previous search's matches, as consecutive ICP iterations are).

The shapes are chosen to hit the search's special paths: degenerate extents (planes, lines: one
or two grid axes collapse), duplicates (ties resolve to the lowest index), cloud sizes that are
not multiples of the 4-wide scan group (the reads past a run's end), far-away and huge
coordinates (box clamping, float cell assignment slack), tiny clouds, dense clumps (long runs)
and queries with nothing within max_corr (full-radius scans on the coarsest level)."""
import numpy as np
import pytest

from libwave_amd import synth

pytestmark = pytest.mark.gpu


def _expect(oracle, ref, tgt, T, max_corr):
    moved = oracle.transform_cloud_f(ref, T.astype(np.float32))
    oi, od = oracle.KdTree(tgt).nn(moved)
    keep = od.astype(np.float64) <= float(max_corr) ** 2
    return np.where(keep, oi, -1), od, keep


def _run(wm, ctx, oracle, ref, tgt, poses, max_corr, method=None):
    method = wm.WM_NN_GRID if method is None else method
    ref = np.ascontiguousarray(ref, np.float32)
    tgt = np.ascontiguousarray(tgt, np.float32)
    ctx.set_source(ref)
    ctx.set_target(tgt)
    for k, T in enumerate(poses):
        flag = method | (wm.WM_NN_WARM if k > 0 else 0)
        gi, gd = ctx.nn_search(T, max_corr, flag)
        wi, wd, keep = _expect(oracle, ref, tgt, T, max_corr)
        bad = gi != wi
        assert not bad.any(), "pose %d: %d wrong indices (first at %d: got %d want %d)" % (
            k, bad.sum(), np.argmax(bad), gi[np.argmax(bad)], wi[np.argmax(bad)])
        assert np.array_equal(gd[keep], wd[keep]), "pose %d: distances not bit-identical" % k


def _poses():
    return [synth.make_T((0.3, -0.2, 0.1), (0.02, -0.01, 0.04)),      # cold
            synth.make_T((0.25, -0.15, 0.08), (0.015, -0.01, 0.03)),  # warm, small step
            synth.make_T((-0.6, 0.4, 0.0), (0.0, 0.05, -0.08)),       # warm, big jump
            np.eye(4),                                                # warm, back to identity
            np.eye(4)]                                                # warm, no motion at all


@pytest.mark.parametrize("n", [1, 2, 3, 5, 7, 63, 64, 65, 257, 1001, 4099])
def test_sizes_not_multiples_of_the_scan_group(wm, ctx, oracle, n):
    rng = np.random.default_rng(n)
    tgt = rng.uniform(-2, 2, (n, 3)).astype(np.float32)
    ref = rng.uniform(-2.5, 2.5, (max(n, 50), 3)).astype(np.float32)
    _run(wm, ctx, oracle, ref, tgt, _poses(), 1.0)


@pytest.mark.parametrize("shape", ["plane", "line", "point", "two_clumps", "shell"])
@pytest.mark.parametrize("max_corr", [0.25, 5.0])
def test_degenerate_and_clumped_targets(wm, ctx, oracle, shape, max_corr):
    rng = np.random.default_rng(7)
    n = 30000
    if shape == "plane":
        tgt = np.c_[rng.uniform(-10, 10, (n, 2)), np.full(n, 1.25)]
    elif shape == "line":
        tgt = np.c_[rng.uniform(-20, 20, n), np.full(n, -3.0), np.full(n, 0.5)]
    elif shape == "point":
        tgt = np.tile(np.array([[1.0, 2.0, 3.0]]), (n, 1))  # n duplicates: index 0 must win
    elif shape == "two_clumps":
        a = rng.normal(0, 0.02, (n // 2, 3)) + [5, 5, 0]
        b = rng.normal(0, 0.02, (n - n // 2, 3)) - [5, 5, 0]
        tgt = np.r_[a, b]
    else:
        v = rng.normal(size=(n, 3))
        tgt = 8.0 * v / np.linalg.norm(v, axis=1, keepdims=True)
    ref = tgt[rng.permutation(n)[:20000]] + rng.normal(0, 0.05, (20000, 3))
    _run(wm, ctx, oracle, ref, tgt, _poses(), max_corr)


def test_duplicates_resolve_to_the_lowest_index(wm, ctx, oracle):
    rng = np.random.default_rng(3)
    base = rng.uniform(-5, 5, (5000, 3)).astype(np.float32)
    tgt = np.r_[base, base[::-1], base]          # every point three times, shuffled order
    ref = base + np.float32(0.01)
    _run(wm, ctx, oracle, ref, tgt, [np.eye(4), np.eye(4), synth.make_T((0.1, 0, 0), (0, 0, 0.01))], 2.0)
    gi, _ = ctx.nn_search(np.eye(4), 2.0, wm.WM_NN_GRID)
    assert gi.max() < 3 * 5000


def test_huge_coordinates_and_far_queries(wm, ctx, oracle):
    rng = np.random.default_rng(5)
    off = np.array([12345.0, -54321.0, 250.0])
    tgt = rng.uniform(-15, 15, (40000, 3)) + off       # UTM-like offsets: ~1 mm float spacing
    ref = np.r_[tgt[:15000] + rng.normal(0, 0.05, (15000, 3)),
                rng.uniform(-15, 15, (2000, 3)) + off + [0, 0, 300.0],   # far above: no match
                rng.uniform(-400, 400, (2000, 3)) + off]                # mostly far outside the box
    _run(wm, ctx, oracle, ref, tgt, [np.eye(4), np.eye(4)], 3.0)


def test_sparse_queries_need_the_full_radius(wm, ctx, oracle):
    """Half of the queries have nothing within max_corr; a quarter sit just inside / outside it."""
    rng = np.random.default_rng(9)
    tgt = np.c_[rng.uniform(-20, 20, (50000, 2)), rng.normal(0, 0.01, 50000)]
    h = np.r_[rng.uniform(0.0, 0.5, 5000), rng.uniform(1.9, 2.1, 5000), rng.uniform(2.5, 30, 10000)]
    ref = np.c_[rng.uniform(-20, 20, (20000, 2)), h]
    _run(wm, ctx, oracle, ref, tgt, _poses(), 2.0)


@pytest.mark.parametrize("cell", [0.03, 0.5, 4.0])
def test_any_cell_size_gives_the_same_answer(wm, ctx, oracle, cell):
    ref, tgt, _ = synth.pair(30000, seed=17, mode="resample")
    ctx.set_grid_cell(cell)
    try:
        _run(wm, ctx, oracle, ref, tgt, _poses(), 3.0)
    finally:
        ctx.set_grid_cell(0.0)


def test_warm_search_after_brute_force(wm, ctx, oracle):
    ref, tgt, _ = synth.pair(3000, seed=23, mode="resample")
    ctx.set_source(ref)
    ctx.set_target(tgt)
    ctx.nn_search(np.eye(4), 3.0, wm.WM_NN_BRUTE)
    T = synth.make_T((0.2, 0.1, 0.0), (0.0, 0.0, 0.02))
    gi, gd = ctx.nn_search(T, 3.0, wm.WM_NN_GRID | wm.WM_NN_WARM)
    wi, wd, keep = _expect(oracle, ref, tgt, T, 3.0)
    assert np.array_equal(gi, wi) and np.array_equal(gd[keep], wd[keep])


def _tiny_steps(rng, n):
    """A converging pose sequence: steps shrink from centimetres to micrometres (what the
    warm searches of a converging ICP see: seeds that are almost always still the answer)."""
    T = synth.make_T((0.05, -0.03, 0.02), (0.004, -0.002, 0.006))
    out = [T]
    for k in range(n):
        s = 0.5 ** k
        dT = synth.make_T(tuple(rng.normal(0, 0.01 * s, 3)), tuple(rng.normal(0, 0.002 * s, 3)))
        T = dT @ T
        out.append(T)
    return out + [out[-1], out[-1]]


def test_converging_warm_sequence_stays_exact(wm, ctx, oracle):
    rng = np.random.default_rng(31)
    ref, tgt, _ = synth.pair(40000, seed=29, mode="resample")
    _run(wm, ctx, oracle, ref, tgt, _tiny_steps(rng, 18), 3.0)


def test_warm_searches_keep_resolving_ties_to_the_lowest_index(wm, ctx, oracle):
    """Target = a regular lattice, queries at (and a hair off) cell centres and face centres:
    2-, 4- and 8-way exact ties in float distance must keep resolving to the lowest index while
    the pose creeps by micrometres."""
    g = np.arange(-8, 9, dtype=np.float32) * np.float32(0.5)
    tgt = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    rng = np.random.default_rng(37)
    tgt = tgt[rng.permutation(len(tgt))]                     # indices unrelated to position
    c = np.stack(np.meshgrid(g[:-1], g[:-1], g[:-1], indexing="ij"), -1).reshape(-1, 3)
    ref = np.r_[c + np.float32(0.25),                        # cell centres: 8-way ties
                c + np.array([0.25, 0.25, 0.0], np.float32),  # face centres: 4-way
                c + np.array([0.25, 0.0, 0.0], np.float32),   # edge centres: 2-way
                c + rng.normal(0, 1e-4, c.shape).astype(np.float32) + np.float32(0.25)]
    poses = [np.eye(4), np.eye(4)]
    for k in range(6):
        poses.append(synth.make_T((1e-6 * (k + 1), -2e-6 * k, 1e-6), (0, 0, 1e-7 * k)))
    poses += [np.eye(4), np.eye(4)]
    _run(wm, ctx, oracle, ref, tgt, poses, 1.0)
