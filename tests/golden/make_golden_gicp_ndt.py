#!/usr/bin/env python3
"""Generate tests/golden/gicp_ndt_golden.json -- INDEPENDENT numpy / scipy restatements of the PCL 1.8
NDT and GICP arithmetic, run on the reference's own fixture (wave_matching/tests/data/testscan.pcd,
committed as tests/golden/testscan.pcd) under the reference tests' own perturbations
(wave_matching/tests/ndt_tests.cpp:45-102, gicp_tests.cpp:43-100).

Independent = no code, table or solver shared with oracle/ (C) or the HIP kernels:
  NDT   voxel model with np.linalg.eigh / inv; neighbours with scipy's cKDTree over the voxel means;
        the point Jacobian and the second derivatives come from PRODUCTS OF ELEMENTARY ROTATION
        MATRICES AND THEIR DERIVATIVES (dRx/da Ry Rz, ...), not from PCL's / Magnusson's tabulated
        trigonometric entries -- so the tables in the oracle and in the kernel are checked against
        calculus, including the one entry PCL has wrong (h_ang d1[2] = +sy, true value -sy): its
        effect on H(4,4) is stated separately;
        the optimum: scipy.optimize.minimize on the negated score, from identity -- where a converged
        NDT (Newton + More-Thuente, either implementation) must land.
  GICP  covariances with cKDTree k-NN + np.linalg.svd; Mahalanobis matrices with np.linalg.inv;
        objective and gradient from rotation-derivative matrix products; the registration's FIXED
        POINT: pair -> minimise (scipy BFGS to 1e-12) -> re-pair, until the pose stops moving.
        PCL's own inner optimiser stops at a gradient tolerance of 1e-2, so on noise-free copies
        (the reference's cases) it ends within ~1e-5 of this fixed point.

The reference itself (PCL) cannot be run in the build image, so these vectors are outputs of this
script, not of libwave; the reference tests' own assertions (|T - T_gt|_F < 0.12 / 0.1) are
re-checked on them.

    python tests/golden/make_golden_gicp_ndt.py        # rewrites gicp_ndt_golden.json (~2 min)
"""
import json
import os
import sys

import numpy as np
from scipy.optimize import minimize
from scipy.spatial import cKDTree

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from libwave_amd.pcd import load_pcd_xyz  # noqa: E402  (file-format reader only)


# ------------------------------------------------------------------ rotations
def elem(axis, a, order=0, small_angle_rule=False):
    """Elementary rotation about `axis` (0 x, 1 y, 2 z) or its `order`-th derivative in the angle.
    small_angle_rule: PCL's computeAngleDerivatives treats |angle| < 10e-5 as cos = 1, sin = 0."""
    c, s = np.cos(a), np.sin(a)
    if small_angle_rule and abs(a) < 10e-5:
        c, s = 1.0, 0.0
    cs = [(c, s), (-s, c), (-c, -s)][order]   # (cos, sin) differentiated `order` times
    cc, ss = cs
    i, j = [(1, 2), (2, 0), (0, 1)][axis]
    R = np.zeros((3, 3))
    if order == 0:
        R[axis, axis] = 1.0
    R[i, i] = cc
    R[j, j] = cc
    R[i, j] = -ss
    R[j, i] = ss
    return R


def transform_d(xyz, T):  # pcl::transformPointCloud(.., Affine3d): double math, float store
    return (xyz.astype(np.float64) @ T[:3, :3].T + T[:3, 3]).astype(np.float32)


# ------------------------------------------------------------------------ NDT
def ndt_voxels(tgt, res):
    """VoxelGridCovariance::applyFilter: leaf = floor(p * inverse_leaf) with a FLOAT inverse leaf,
    >= 6 points, single-pass covariance with PCL's (n-1)/n factor, eigenvalue floor at 1 % of the
    largest, inverse covariance."""
    inv = np.float32(1.0) / np.float32(res)
    ijk = np.floor(tgt * inv).astype(np.int64)
    order = np.lexsort((ijk[:, 0], ijk[:, 1], ijk[:, 2]))
    s = ijk[order]
    brk = np.flatnonzero(np.r_[True, np.any(s[1:] != s[:-1], axis=1)])
    cnt = np.diff(np.r_[brk, len(s)])
    out = []
    P = tgt[order].astype(np.float64)
    for b, n in zip(brk, cnt):
        if n < 6:
            continue
        p = P[b:b + n]
        sm = p.sum(0)
        mean = sm / n
        cov = (p.T @ p - 2.0 * np.outer(sm, mean)) / n + np.outer(mean, mean)
        cov *= (n - 1.0) / n
        w, V = np.linalg.eigh(cov)
        if w[0] < 0 or w[1] < 0 or w[2] <= 0:
            continue
        lo = 0.01 * w[2]
        if w[0] < lo:
            w = np.array([lo, max(w[1], lo), w[2]])
            cov = V @ np.diag(w) @ np.linalg.inv(V)
        icov = np.linalg.inv(cov)
        if not np.all(np.isfinite(icov)):
            continue
        out.append((tuple(int(v) for v in s[b]), int(n), mean, icov))
    return out


def ndt_constants(res, outlier_ratio=0.55):
    c1 = 10.0 * (1 - outlier_ratio)
    c2 = outlier_ratio / res ** 3
    d3 = -np.log(c2)
    d1 = -np.log(c1 + c2) - d3
    d2 = -2 * np.log((-np.log(c1 * np.exp(-0.5) + c2) - d3) / d1)
    return d1, d2


def ndt_pose_matrix(p):  # Translation * Rx * Ry * Rz
    T = np.eye(4)
    T[:3, :3] = elem(0, p[3]) @ elem(1, p[4]) @ elem(2, p[5])
    T[:3, 3] = p[:3]
    return T


def ndt_eval(vox, src, p, res, hessian=True):
    """score, gradient, Hessian of PCL's NDT objective at pose p = (t, rx, ry, rz); also the change
    of H[4,4] that PCL's h_ang d1[2] = +sy (instead of -sy) makes."""
    d1, d2 = ndt_constants(res)
    means = np.array([v[2] for v in vox])
    icovs = np.array([v[3] for v in vox])
    tree = cKDTree(means)
    Tf = ndt_pose_matrix(p).astype(np.float32)     # PCL moves the cloud with a float matrix
    x = src.astype(np.float32)
    xt = np.stack([((Tf[r, 0] * x[:, 0] + Tf[r, 1] * x[:, 1]) + Tf[r, 2] * x[:, 2]) + Tf[r, 3] for r in range(3)],
                  axis=1).astype(np.float64)
    # derivative factors of R = Rx Ry Rz with PCL's small-angle rule
    A = [[elem(ax, p[3 + ax], o, True) for o in range(3)] for ax in range(3)]

    def Rd(oa, ob, oc):
        return A[0][oa] @ A[1][ob] @ A[2][oc]
    dR = [Rd(1, 0, 0), Rd(0, 1, 0), Rd(0, 0, 1)]
    d2R = {(0, 0): Rd(2, 0, 0), (0, 1): Rd(1, 1, 0), (0, 2): Rd(1, 0, 1),
           (1, 1): Rd(0, 2, 0), (1, 2): Rd(0, 1, 1), (2, 2): Rd(0, 0, 2)}
    sy = 0.0 if abs(p[4]) < 10e-5 else np.sin(p[4])
    nb = tree.query_ball_point(xt, r=res)
    I = np.repeat(np.arange(len(xt)), [len(v) for v in nb])
    K = np.fromiter((k for v in nb for k in v), dtype=np.int64, count=len(I))
    xo = src.astype(np.float64)[I]
    xc = xt[I] - means[K]
    C = icovs[K]
    Cx = np.einsum("pij,pj->pi", C, xc)            # S^-1 x (S^-1 symmetric up to rounding)
    q = (xc * Cx).sum(1)
    e = np.exp(-d2 * q / 2)
    score = float((-d1 * e).sum())
    w = d2 * e
    w = np.where((w <= 1) & (w >= 0), w * d1, 0.0)  # PCL drops terms whose d2 e leaves [0, 1]
    J = np.zeros((len(I), 3, 6))
    J[:, :, :3] = np.eye(3)
    for a in range(3):
        J[:, :, 3 + a] = xo @ dR[a].T
    cj = np.einsum("pi,pia->pa", Cx, J)
    g = (w[:, None] * cj).sum(0)
    H = np.zeros((6, 6))
    typo44 = 0.0
    if hessian:
        H = -d2 * np.einsum("p,pa,pb->ab", w, cj, cj) + np.einsum("p,pia,pij,pjb->ab", w, J, C, J)
        for (a, b), D in d2R.items():
            t = float((w * (Cx * (xo @ D.T)).sum(1)).sum())
            H[3 + a, 3 + b] += t
            if a != b:
                H[3 + b, 3 + a] += t
        typo44 = float((w * Cx[:, 0] * 2.0 * sy * xo[:, 2]).sum())
    return score, g, H, typo44


def ndt_optimum(vox, src, res):
    f = lambda p: tuple(np.negative(v) for v in ndt_eval(vox, src, p, res, hessian=False)[:2])  # noqa: E731
    r = minimize(lambda p: f(p)[0], np.zeros(6), jac=lambda p: f(p)[1], method="BFGS",
                 options=dict(gtol=1e-7, maxiter=200))
    return r.x, -r.fun


# ----------------------------------------------------------------------- GICP
def gicp_covariances(xyz, k=10, eps=1e-3):
    """computeCovariances: k nearest neighbours (self included), mean and covariance with PCL's FLOAT
    products accumulated in double, SVD, singular values replaced by (1, 1, eps)."""
    tree = cKDTree(xyz.astype(np.float64))
    _, nn = tree.query(xyz.astype(np.float64), k=k)
    P = xyz[nn]                                   # (n, k, 3) float32
    mean = P.astype(np.float64).sum(1) / k
    prod = (P[:, :, :, None] * P[:, :, None, :]).astype(np.float64).sum(1) / k   # float32 products
    cov = prod - mean[:, :, None] * mean[:, None, :]
    cov = np.tril(cov) + np.tril(cov, -1).transpose(0, 2, 1)   # PCL fills the lower triangle and mirrors it
    U, _, _ = np.linalg.svd(cov)
    D = np.diag([1.0, 1.0, eps])
    return U @ D @ U.transpose(0, 2, 1), nn


def gicp_matrix_f(x):  # applyState on identity: Rz(psi) Ry(theta) Rx(phi), float
    xf = np.asarray(x, np.float64)
    R = (elem(2, xf[5]) @ elem(1, xf[4]) @ elem(0, xf[3])).astype(np.float32)
    T = np.eye(4, dtype=np.float32)
    T[:3, :3] = R
    T[:3, 3] = xf[:3].astype(np.float32)
    return T


def gicp_fdf(src, tgt, si, ti, M, x):
    """f = (1/m) sum r^T M r, r = T(x) p - q (float transform, as PCL), and its gradient."""
    Tf = gicp_matrix_f(x)
    p = src[si]
    pp = np.stack([((Tf[r, 0] * p[:, 0] + Tf[r, 1] * p[:, 1]) + Tf[r, 2] * p[:, 2]) + Tf[r, 3] for r in range(3)], axis=1)
    r = (pp - tgt[ti]).astype(np.float64)
    Mr = np.einsum("nij,nj->ni", M[si], r)
    m = len(si)
    f = float((r * Mr).sum() / m)
    g = np.zeros(6)
    g[:3] = 2.0 * Mr.sum(0) / m
    Racc = 2.0 * (p.astype(np.float64).T @ Mr) / m          # sum p (M r)^T
    dR = [elem(2, x[5]) @ elem(1, x[4]) @ elem(0, x[3], 1), elem(2, x[5]) @ elem(1, x[4], 1) @ elem(0, x[3]),
          elem(2, x[5], 1) @ elem(1, x[4]) @ elem(0, x[3])]
    for a in range(3):
        g[3 + a] = float((dR[a] * Racc.T).sum())            # tr(dR^T ... ) = sum_ab dR_ab A_ba
    return f, g


def gicp_fixed_point(src, tgt, k=10, eps=1e-3, max_corr=5.0, outer=30):
    C1, _ = gicp_covariances(src, k, eps)
    C2, _ = gicp_covariances(tgt, k, eps)
    tree = cKDTree(tgt.astype(np.float64))
    x = np.zeros(6)
    hist = []
    for _ in range(outer):
        Tf = gicp_matrix_f(x)
        moved = np.stack([((Tf[r, 0] * src[:, 0] + Tf[r, 1] * src[:, 1]) + Tf[r, 2] * src[:, 2]) + Tf[r, 3]
                          for r in range(3)], axis=1)
        d, j = tree.query(moved.astype(np.float64))
        diff = moved - tgt[j]
        d2 = (diff[:, 0] * diff[:, 0] + diff[:, 1] * diff[:, 1]) + diff[:, 2] * diff[:, 2]
        keep = d2.astype(np.float64) < max_corr * max_corr
        si = np.flatnonzero(keep)
        ti = j[si]
        R = Tf[:3, :3].astype(np.float64)
        M = np.zeros((len(src), 3, 3))
        M[si] = np.linalg.inv(C2[ti] + R @ C1[si] @ R.T)
        r = minimize(lambda y: gicp_fdf(src, tgt, si, ti, M, y)[0], x, jac=lambda y: gicp_fdf(src, tgt, si, ti, M, y)[1],
                     method="BFGS", options=dict(gtol=1e-12, maxiter=400))
        step = float(np.abs(r.x - x).max())
        x = r.x
        hist.append(dict(pairs=int(len(si)), f=float(r.fun), step=step))
        if step < 1e-9:
            break
    return gicp_matrix_f(x).astype(np.float64), x, hist, C1, C2


# ----------------------------------------------------------------------- main
def main():
    scan = load_pcd_xyz(os.path.join(HERE, "testscan.pcd"))
    out = {"fixture_sha256": "c22245b9eb63abd8537a38d0704337f659ff973ae5e0e5fa7ef89a7d21cdaaa8",
           "ndt": {}, "gicp": {}}
    # ---- NDT: the reference's three cases (res 0.05 / 0.1 identity, res 0.3 with +0.2 m)
    poses = [np.zeros(6), np.array([0.05, -0.02, 0.01, 0.004, -0.003, 0.006])]
    for name, res, tx in (("nullDisplacement", 0.1, 0.0), ("smallDisplacement", 0.3, 0.2), ("coarse", 1.0, 0.2)):
        P = np.eye(4)
        P[0, 3] = tx
        target = transform_d(scan, P)
        vox = ndt_voxels(target, res)
        src = scan[::4] if res < 0.3 else scan      # (the fine grids: every 4th point keeps this script quick)
        c = {"res": res, "tx": tx, "source_stride": 4 if res < 0.3 else 1, "n_voxels": len(vox),
             "first_voxels": [dict(ijk=v[0], n=v[1], mean=v[2].tolist(), icov=v[3].tolist()) for v in vox[:6]],
             "evals": []}
        for p in poses:
            s, g, H, typo = ndt_eval(vox, src, p, res)
            c["evals"].append(dict(pose=p.tolist(), score=s, grad=g.tolist(), hess=H.tolist(),
                                   hess44_pcl_minus_true=typo))
        if name == "smallDisplacement":
            # PCL 1.8.x's step rule (its More-Thuente loop never runs): the undamped Newton
            # step dp = H^-1 (-g) -- H with PCL's own h_ang d1 entry --, flipped when it is not an ascent
            # direction of the score, its length clamped to step_size = 3.  The first steps from p = 0
            # (ndt_tests.cpp:85-102); later ones are chaotic and pin nothing.
            pk = np.zeros(6)
            steps = []
            for _ in range(4):
                s_, g_, H_, typo_ = ndt_eval(vox, src, pk, res)
                Hp = H_.copy()
                Hp[4, 4] += typo_
                dp = np.linalg.solve(Hp, -g_)
                nrm = float(np.linalg.norm(dp))
                d = dp / nrm
                if -(g_ @ d) > 0:          # d_phi_0 = -(g . dir) >= 0: not a descent direction of -score
                    d = -d
                pk = pk + d * min(nrm, 3.0)
                steps.append(dict(pose=pk.tolist(), T=ndt_pose_matrix(pk).tolist(), newton_norm=nrm))
            c["pcl18_newton_steps"] = steps
        if res >= 0.3:
            x, sc = ndt_optimum(vox, src, res)
            c["optimum_pose"] = x.tolist()
            c["optimum_score"] = sc
            c["optimum_T"] = ndt_pose_matrix(x).tolist()
            frob = float(np.linalg.norm(ndt_pose_matrix(x) - P))
            assert frob < 0.12, (name, frob)        # the reference test's own assertion
            c["frob_vs_gt"] = frob
        out["ndt"][name] = c
        print("ndt", name, c["n_voxels"], c["evals"][0]["score"], c.get("optimum_pose"))
    # ---- GICP: the reference's smallDisplacement case on the voxel-filtered clouds (res 0.05 is what
    # gicp_tests.cpp uses; the filter itself is pinned in icp_golden.json) and a full-resolution one
    from make_golden import voxel_grid              # the independent VoxelGrid restatement (same directory)
    for name, res, tx in (("smallDisplacement", 0.05, 0.2), ("fullResSmallDisplacement", -1.0, 0.2)):
        P = np.eye(4)
        P[0, 3] = tx
        target = transform_d(scan, P)
        a = scan if res < 0 else voxel_grid(scan, res)
        b = target if res < 0 else voxel_grid(target, res)
        T, x, hist, C1, C2 = gicp_fixed_point(a, b)
        frob = float(np.linalg.norm(T - P))
        assert frob < 0.1, (name, frob)             # gicp_tests.cpp:36
        probe = np.array([0.12, -0.06, 0.03, 0.005, -0.012, 0.017])
        tree = cKDTree(b.astype(np.float64))
        d, j = tree.query(a.astype(np.float64))
        si = np.arange(len(a))
        M = np.linalg.inv(C2[j] + C1)
        f0, g0 = gicp_fdf(a, b, si, j, M, probe)
        out["gicp"][name] = {"res": res, "tx": tx, "n_ref": int(len(a)), "n_target": int(len(b)),
                             "fixed_point_T": T.tolist(), "fixed_point_x": x.tolist(), "outer": hist,
                             "frob_vs_gt": frob,
                             "cov_ref_first4": C1[:4].tolist(), "cov_target_first4": C2[:4].tolist(),
                             "cov_ref_checksum": float(np.abs(C1).sum()), "cov_target_checksum": float(np.abs(C2).sum()),
                             "probe": dict(x=probe.tolist(), f=f0, grad=g0.tolist(),
                                           note="pairs = nearest neighbour under the identity, M = inv(C2[j] + C1[i])")}
        print("gicp", name, len(a), frob, hist[-1])
    # ---- GICP with the voxel filter on a NOISY pair: the scan against a re-measured copy (1 cm of
    # Gaussian noise per coordinate, fixed seed) under a small rigid motion -- residuals do not vanish
    # at the optimum, which is where PCL's early-stopping BFGS and an exact minimiser part ways (mm)
    rng = np.random.Generator(np.random.PCG64(20240917))
    Pn = np.eye(4)
    cz, sz = np.cos(0.01), np.sin(0.01)
    Pn[:3, :3] = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    Pn[:3, 3] = [0.15, -0.05, 0.02]
    noisy = (transform_d(scan, Pn).astype(np.float64) + rng.normal(0.0, 0.01, scan.shape)).astype(np.float32)
    a, b = voxel_grid(scan, 0.1), voxel_grid(noisy, 0.1)
    T, x, hist, C1, C2 = gicp_fixed_point(a, b)
    out["gicp"]["noisyFiltered"] = {"res": 0.1, "P": Pn.tolist(), "noise_sigma": 0.01, "noise_seed": 20240917,
                                    "n_ref": int(len(a)), "n_target": int(len(b)), "fixed_point_T": T.tolist(),
                                    "fixed_point_x": x.tolist(), "outer": hist,
                                    "frob_vs_gt": float(np.linalg.norm(T - Pn)),
                                    "target_checksum": float(np.abs(noisy.astype(np.float64)).sum())}
    print("gicp noisyFiltered", len(a), len(b), out["gicp"]["noisyFiltered"]["frob_vs_gt"], hist[-1])
    with open(os.path.join(HERE, "gicp_ndt_golden.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
