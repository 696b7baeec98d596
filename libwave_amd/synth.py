"""Deterministic synthetic scenes for BASELINE.json's configs.

Scene S(n, seed): 40 % ground (x in [-50,50], y in [-30,30], z ~ N(0, 0.01^2)),
30 % on the four walls of the 100 x 60 m box (z in [0, 8]), 30 % on 200 random
boxes (<= 4 x 2 x 2 m).  float32 XYZ.  Pair modes: 'copy' (target = T_gt * ref,
mirrors wave_matching/tests/icp_tests.cpp:31) and 'resample' (target = T_gt *
S(n, seed+1) + N(0, 0.01^2)) so the MSE floor is non-zero.
"""
import numpy as np


def rpy_to_R(roll, pitch, yaw):
    cr, sr, cp, sp, cy, sy = (np.cos(roll), np.sin(roll), np.cos(pitch), np.sin(pitch),
                              np.cos(yaw), np.sin(yaw))
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def make_T(t=(0.2, -0.1, 0.05), rpy=(0.01, -0.02, 0.03)):
    T = np.eye(4)
    T[:3, :3] = rpy_to_R(*rpy)
    T[:3, 3] = t
    return T


T_GT = make_T()


def scene(n, seed=42, layout_seed=20240):
    """n points sampled (seed) on a fixed scene geometry (layout_seed)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    n_ground = int(0.4 * n)
    n_wall = int(0.3 * n)
    n_box = n - n_ground - n_wall
    g = np.empty((n_ground, 3))
    g[:, 0] = rng.uniform(-50, 50, n_ground)
    g[:, 1] = rng.uniform(-30, 30, n_ground)
    g[:, 2] = rng.normal(0, 0.01, n_ground)
    w = np.empty((n_wall, 3))
    side = rng.integers(0, 4, n_wall)
    u = rng.uniform(0, 1, n_wall)
    w[:, 2] = rng.uniform(0, 8, n_wall)
    w[:, 0] = np.where(side == 0, -50, np.where(side == 1, 50, -50 + 100 * u))
    w[:, 1] = np.where(side == 2, -30, np.where(side == 3, 30, -30 + 60 * u))
    # 200 boxes; the layout is part of the scene geometry, NOT of the sampling seed,
    # so S(n, seed) and S(n, seed + 1) are two samplings of the same surfaces
    brng = np.random.Generator(np.random.PCG64(layout_seed))
    nb = 200
    centre = np.stack([brng.uniform(-45, 45, nb), brng.uniform(-25, 25, nb)], axis=1)
    size = np.stack([brng.uniform(0.5, 4, nb), brng.uniform(0.5, 2, nb), brng.uniform(0.5, 2, nb)], axis=1)
    which = rng.integers(0, nb, n_box)
    face = rng.integers(0, 5, n_box)  # 4 sides + top
    a = rng.uniform(-0.5, 0.5, n_box)
    b = rng.uniform(-0.5, 0.5, n_box)
    sx, sy, sz = size[which, 0], size[which, 1], size[which, 2]
    bx = np.where(face == 0, -0.5 * sx, np.where(face == 1, 0.5 * sx, a * sx))
    by = np.where(face == 2, -0.5 * sy, np.where(face == 3, 0.5 * sy, np.where(face < 2, a * sy, b * sy)))
    bz = np.where(face == 4, sz, (b + 0.5) * sz)
    bpts = np.stack([centre[which, 0] + bx, centre[which, 1] + by, bz], axis=1)
    pts = np.concatenate([g, w, bpts], axis=0)
    pts = pts[rng.permutation(len(pts))]
    return np.ascontiguousarray(pts, dtype=np.float32)


def scene_rings(n, seed=42, layout_seed=20240, rings=64, sensor_z=1.73):
    """The same scene seen the way a spinning 64-ring lidar sees it ("KITTI-like": BASELINE
    configs[3]): 85 % of the points are ray hits from a sensor at (0, 0, sensor_z)
    -- 64 elevation angles from -24.8 to +2.0 degrees (an HDL-64E's fan), uniform azimuth -- on
    the ground and, where the ray leaves the 100 x 60 m footprint first, on the walls: concentric
    ground rings, dense near the sensor and sparse far out, and stacked arcs on the walls.  15 %
    are on the 200 boxes (sampled as in scene(); occlusion is not modelled).  n points exactly."""
    rng = np.random.Generator(np.random.PCG64(seed))
    n_box = int(0.15 * n)
    n_ray = n - n_box
    elev = np.deg2rad(np.linspace(-24.8, 2.0, rings))
    out = np.empty((0, 3))
    while len(out) < n_ray:
        m = int((n_ray - len(out)) * 1.4) + 1024
        th = elev[rng.integers(0, rings, m)]
        ph = rng.uniform(0.0, 2.0 * np.pi, m)
        d = np.stack([np.cos(th) * np.cos(ph), np.cos(th) * np.sin(ph), np.sin(th)], axis=1)
        with np.errstate(divide="ignore", invalid="ignore"):
            t_g = np.where(d[:, 2] < 0, -sensor_z / d[:, 2], np.inf)
            t_x = np.where(d[:, 0] > 0, 50.0 / d[:, 0], np.where(d[:, 0] < 0, -50.0 / d[:, 0], np.inf))
            t_y = np.where(d[:, 1] > 0, 30.0 / d[:, 1], np.where(d[:, 1] < 0, -30.0 / d[:, 1], np.inf))
        t_w = np.minimum(t_x, t_y)
        on_ground = t_g < t_w
        t = np.where(on_ground, t_g, t_w) * (1.0 + rng.normal(0.0, 2e-4, m))  # range jitter
        p = d * t[:, None]
        p[:, 2] += sensor_z
        p[on_ground, 2] = rng.normal(0.0, 0.01, int(on_ground.sum()))
        keep = on_ground | ((p[:, 2] >= 0.0) & (p[:, 2] <= 8.0))
        out = np.concatenate([out, p[keep]], axis=0)
    out = out[:n_ray]
    brng = np.random.Generator(np.random.PCG64(layout_seed))
    nb = 200
    centre = np.stack([brng.uniform(-45, 45, nb), brng.uniform(-25, 25, nb)], axis=1)
    size = np.stack([brng.uniform(0.5, 4, nb), brng.uniform(0.5, 2, nb), brng.uniform(0.5, 2, nb)], axis=1)
    which = rng.integers(0, nb, n_box)
    face = rng.integers(0, 5, n_box)
    a = rng.uniform(-0.5, 0.5, n_box)
    b = rng.uniform(-0.5, 0.5, n_box)
    sx, sy, sz = size[which, 0], size[which, 1], size[which, 2]
    bx = np.where(face == 0, -0.5 * sx, np.where(face == 1, 0.5 * sx, a * sx))
    by = np.where(face == 2, -0.5 * sy, np.where(face == 3, 0.5 * sy, np.where(face < 2, a * sy, b * sy)))
    bz = np.where(face == 4, sz, (b + 0.5) * sz)
    bpts = np.stack([centre[which, 0] + bx, centre[which, 1] + by, bz], axis=1)
    pts = np.concatenate([out, bpts], axis=0)
    pts = pts[rng.permutation(len(pts))]
    return np.ascontiguousarray(pts, dtype=np.float32)


def transform_points(xyz, T):
    """pcl::transformPointCloud(in, out, Affine3d): double arithmetic, float store."""
    p = xyz.astype(np.float64)
    return np.ascontiguousarray((p @ T[:3, :3].T + T[:3, 3]).astype(np.float32))


def pair(n, seed=42, mode="resample", T=None, noise=0.01, pattern="uniform"):
    """Returns (ref, target, T_gt) with target ~= T_gt * ref.  pattern: "uniform" (area-uniform
    sampling of the surfaces, scene()) or "rings" (a 64-ring lidar's sampling, scene_rings())."""
    T = T_GT if T is None else T
    gen = scene if pattern == "uniform" else scene_rings
    ref = gen(n, seed)
    if mode == "copy":
        tgt = transform_points(ref, T)
    elif mode == "resample":
        other = gen(n, seed + 1).astype(np.float64)
        rng = np.random.Generator(np.random.PCG64(seed + 1000003))
        other = other + rng.normal(0, noise, other.shape)
        tgt = transform_points(other.astype(np.float32), T)
    else:
        raise ValueError(mode)
    return ref, tgt, T


TILE_X = 100.0  # the base scene spans x in [-50, 50]


def pair_tiled(n_per_tile, tiles, seed=42, noise=0.01):
    """Weak-scaling workload: `tiles` copies of the base scene side by side along x (each its
    own sampling), i.e. a map `tiles` times longer at the SAME point density, so that the work
    per point -- which for a radius-bounded search grows with density -- is the same at every
    size.  The ground-truth rotation is divided by `tiles` so that the initial misalignment
    field (<= ~1.7 m at the far ends) is the same too; the translation is unchanged.
    tiles == 1 is exactly pair(n_per_tile, seed, 'resample')."""
    if tiles == 1:
        return pair(n_per_tile, seed=seed, mode="resample", noise=noise)
    T = make_T((0.2, -0.1, 0.05), tuple(a / tiles for a in (0.01, -0.02, 0.03)))
    refs, tgts = [], []
    for k in range(tiles):
        off = np.array([(k - (tiles - 1) / 2.0) * TILE_X, 0.0, 0.0])
        s = seed + 7 * k
        refs.append((scene(n_per_tile, s).astype(np.float64) + off).astype(np.float32))
        other = scene(n_per_tile, s + 1).astype(np.float64) + off
        rng = np.random.Generator(np.random.PCG64(s + 1000003))
        tgts.append((other + rng.normal(0, noise, other.shape)).astype(np.float32))
    ref = np.ascontiguousarray(np.concatenate(refs))
    tgt = transform_points(np.concatenate(tgts), T)
    return ref, tgt, T
