"""ctypes binding of oracle/libwm_oracle.so -- TEST INFRASTRUCTURE ONLY.

The oracle is the CPU restatement of the PCL algorithms libwave's matchers call
(see oracle/wm_oracle.h).  Importers: tests/, __graft_entry__.smoke(), and the
cpu_baseline leg of bench.py.  Never the product path.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "libwm_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    stale = (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _declare(_LIB)
    return _LIB


class IcpParams(C.Structure):
    _fields_ = [("max_corr", C.c_double), ("max_iter", C.c_int), ("t_eps", C.c_double),
                ("fit_eps", C.c_double), ("force_iterations", C.c_int), ("mode", C.c_int),
                ("float_sums", C.c_int), ("incremental_float", C.c_int),
                ("prev_mse_in", C.c_double)]


class IcpResult(C.Structure):
    _fields_ = [("converged", C.c_int), ("iterations", C.c_int), ("state", C.c_int),
                ("n_corr", C.c_int), ("mse", C.c_double), ("prev_mse_out", C.c_double)]


class GicpParams(C.Structure):
    _fields_ = [("corr_rand", C.c_int), ("max_iter", C.c_int), ("r_eps", C.c_double),
                ("t_eps", C.c_double), ("max_corr", C.c_double), ("gicp_epsilon", C.c_double),
                ("max_inner", C.c_int), ("force_iterations", C.c_int)]


class GicpResult(C.Structure):
    _fields_ = [("converged", C.c_int), ("iterations", C.c_int), ("n_corr", C.c_int),
                ("inner_total", C.c_int), ("f_final", C.c_double), ("evaluations", C.c_int),
                ("reserved", C.c_int)]


class NdtParams(C.Structure):
    _fields_ = [("res", C.c_double), ("step_size", C.c_double), ("t_eps", C.c_double),
                ("max_iter", C.c_int), ("outlier_ratio", C.c_double),
                ("skip_line_search", C.c_int), ("pcl_d1_sign", C.c_int),
                ("force_iterations", C.c_int)]


class NdtResult(C.Structure):
    _fields_ = [("converged", C.c_int), ("iterations", C.c_int), ("n_voxels", C.c_int),
                ("score", C.c_double)]


CONV_NAMES = {0: "NOT_CONVERGED", 1: "ITERATIONS", 2: "TRANSFORM", 3: "ABS_MSE", 4: "REL_MSE",
              5: "NO_CORRESPONDENCES", 6: "FORCED"}

_fp = C.POINTER(C.c_float)
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def _declare(L):
    L.wmo_kdtree_build.restype = C.c_void_p
    L.wmo_kdtree_build.argtypes = [_fp, C.c_int]
    L.wmo_kdtree_free.argtypes = [C.c_void_p]
    L.wmo_nn_batch.argtypes = [C.c_void_p, _fp, C.c_int, _ip, _fp]
    L.wmo_kdtree_knn.restype = C.c_int
    L.wmo_kdtree_knn.argtypes = [C.c_void_p, _fp, C.c_int, _ip, _fp]
    L.wmo_nn_brute.argtypes = [_fp, C.c_int, _fp, C.c_int, _ip, _fp]
    L.wmo_transform_cloud_d.argtypes = [_fp, C.c_int, _dp, _fp]
    L.wmo_transform_cloud_f.argtypes = [_fp, C.c_int, _fp, _fp]
    L.wmo_voxel_grid.restype = C.c_int
    L.wmo_voxel_grid.argtypes = [_fp, C.c_int, C.c_float, _fp]
    L.wmo_icp_default_params.argtypes = [C.POINTER(IcpParams)]
    L.wmo_icp_align.restype = C.c_int
    L.wmo_icp_align.argtypes = [_fp, C.c_int, _fp, C.c_int, C.POINTER(IcpParams), _dp,
                                C.POINTER(IcpResult), _ip, _fp, _fp, _dp]
    L.wmo_icp_match.restype = C.c_int
    L.wmo_icp_match.argtypes = [_fp, C.c_int, _fp, C.c_int, C.POINTER(IcpParams), C.c_float,
                                C.c_int, _dp, C.POINTER(IcpResult), C.POINTER(C.c_void_p)]
    L.wmo_match_free.argtypes = [C.c_void_p]
    L.wmo_match_counts.restype = C.c_int
    L.wmo_match_counts.argtypes = [C.c_void_p, _ip, _ip, _ip]
    for name in ("wmo_match_ref", "wmo_match_target", "wmo_match_final"):
        getattr(L, name).restype = _fp
        getattr(L, name).argtypes = [C.c_void_p]
    L.wmo_match_corr.restype = _ip
    L.wmo_match_corr.argtypes = [C.c_void_p]
    L.wmo_info_lum.restype = C.c_int
    L.wmo_info_lum.argtypes = [C.c_void_p, _dp]
    L.wmo_info_lumold.restype = C.c_int
    L.wmo_info_lumold.argtypes = [C.c_void_p, C.c_double, _dp]
    L.wmo_info_censi.restype = C.c_int
    L.wmo_info_censi.argtypes = [C.c_void_p, _dp, C.c_double, C.c_double, _dp]
    L.wmo_lum_from_pairs.restype = C.c_int
    L.wmo_lum_from_pairs.argtypes = [_fp, _fp, C.c_int, _dp, _dp, _dp, _fp]
    L.wmo_censi_from_pairs.restype = C.c_int
    L.wmo_censi_from_pairs.argtypes = [_fp, _fp, C.c_int, _dp, C.c_double, C.c_double, _dp, _dp,
                                       _dp]
    L.wmo_svd.argtypes = [C.c_int, _dp, _dp, _dp, _dp]
    L.wmo_sym_eig.argtypes = [C.c_int, _dp, _dp, _dp]
    L.wmo_inverse.restype = C.c_int
    L.wmo_inverse.argtypes = [C.c_int, _dp, _dp]
    L.wmo_umeyama.argtypes = [_fp, _fp, C.c_int, C.c_int, _dp]
    L.wmo_euler_angles_012.argtypes = [_dp, _dp]
    if hasattr(L, "wmo_gicp_align"):
        L.wmo_gicp_default_params.argtypes = [C.POINTER(GicpParams)]
        L.wmo_gicp_covariances.restype = C.c_int
        L.wmo_gicp_covariances.argtypes = [_fp, C.c_int, C.c_int, C.c_double, _dp]
        L.wmo_gicp_align.restype = C.c_int
        L.wmo_gicp_align.argtypes = [_fp, C.c_int, _fp, C.c_int, C.POINTER(GicpParams), _dp,
                                     C.POINTER(GicpResult)]
        L.wmo_gicp_fdf.restype = C.c_double
        L.wmo_gicp_fdf.argtypes = [_fp, _fp, _ip, _ip, _dp, C.c_int, _dp, _dp, _dp]
    if hasattr(L, "wmo_ndt_align"):
        L.wmo_ndt_default_params.argtypes = [C.POINTER(NdtParams)]
        L.wmo_ndt_grid_build.restype = C.c_void_p
        L.wmo_ndt_grid_build.argtypes = [_fp, C.c_int, C.c_double]
        L.wmo_ndt_grid_free.argtypes = [C.c_void_p]
        L.wmo_ndt_grid_size.restype = C.c_int
        L.wmo_ndt_grid_size.argtypes = [C.c_void_p]
        L.wmo_ndt_grid_export.argtypes = [C.c_void_p, _ip, _dp, _dp, _ip]
        L.wmo_ndt_derivatives.restype = C.c_double
        L.wmo_ndt_derivatives.argtypes = [C.c_void_p, _fp, C.c_int, C.POINTER(NdtParams), _dp,
                                          _dp, _dp]
        L.wmo_ndt_align.restype = C.c_int
        L.wmo_ndt_align.argtypes = [_fp, C.c_int, _fp, C.c_int, C.POINTER(NdtParams), _dp,
                                    C.POINTER(NdtResult)]


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    assert a.ndim == 2 and a.shape[1] == 3
    return a


def _p(a, t):
    return a.ctypes.data_as(t)


# ------------------------------------------------------------------ wrappers
class KdTree:
    def __init__(self, xyz):
        self.xyz = _f32(xyz)
        self.h = lib().wmo_kdtree_build(_p(self.xyz, _fp), len(self.xyz))

    def __del__(self):
        if getattr(self, "h", None):
            lib().wmo_kdtree_free(self.h)
            self.h = None

    def nn(self, q):
        q = _f32(q)
        idx = np.empty(len(q), np.int32)
        d2 = np.empty(len(q), np.float32)
        lib().wmo_nn_batch(self.h, _p(q, _fp), len(q), _p(idx, _ip), _p(d2, _fp))
        return idx, d2

    def knn(self, q, k):
        q = _f32(q)
        idx = np.empty((len(q), k), np.int32)
        d2 = np.empty((len(q), k), np.float32)
        for i in range(len(q)):
            lib().wmo_kdtree_knn(self.h, _p(q[i:i + 1], _fp), k, _p(idx[i], _ip), _p(d2[i], _fp))
        return idx, d2


def nn_brute(tgt, q):
    tgt, q = _f32(tgt), _f32(q)
    idx = np.empty(len(q), np.int32)
    d2 = np.empty(len(q), np.float32)
    lib().wmo_nn_brute(_p(tgt, _fp), len(tgt), _p(q, _fp), len(q), _p(idx, _ip), _p(d2, _fp))
    return idx, d2


def transform_cloud_d(xyz, T):
    xyz = _f32(xyz)
    T = np.ascontiguousarray(T, np.float64)
    out = np.empty_like(xyz)
    lib().wmo_transform_cloud_d(_p(xyz, _fp), len(xyz), _p(T, _dp), _p(out, _fp))
    return out


def transform_cloud_f(xyz, T):
    xyz = _f32(xyz)
    T = np.ascontiguousarray(T, np.float32)
    out = np.empty_like(xyz)
    lib().wmo_transform_cloud_f(_p(xyz, _fp), len(xyz), _p(T, _fp), _p(out, _fp))
    return out


def voxel_grid(xyz, leaf):
    xyz = _f32(xyz)
    out = np.empty_like(xyz)
    n = lib().wmo_voxel_grid(_p(xyz, _fp), len(xyz), C.c_float(leaf), _p(out, _fp))
    return out[:n].copy()


def icp_params(**kw):
    p = IcpParams()
    lib().wmo_icp_default_params(C.byref(p))
    for k, v in kw.items():
        assert hasattr(p, k), k
        setattr(p, k, v)
    return p


def icp_align(src, tgt, params=None, want_corr=False, want_final=False, want_trace=False, **kw):
    src, tgt = _f32(src), _f32(tgt)
    p = params or icp_params(**kw)
    T = np.zeros((4, 4), np.float64)
    r = IcpResult()
    n = len(src)
    idx = np.empty(n, np.int32) if want_corr else None
    d2 = np.empty(n, np.float32) if want_corr else None
    fin = np.empty((n, 3), np.float32) if want_final else None
    cap = max(p.max_iter, p.force_iterations, 1)
    tr = np.zeros((cap, 2), np.float64) if want_trace else None
    rc = lib().wmo_icp_align(_p(src, _fp), n, _p(tgt, _fp), len(tgt), C.byref(p), _p(T, _dp),
                             C.byref(r), _p(idx, _ip) if want_corr else None,
                             _p(d2, _fp) if want_corr else None,
                             _p(fin, _fp) if want_final else None,
                             _p(tr, _dp) if want_trace else None)
    out = dict(rc=rc, T=T, converged=bool(r.converged), iterations=r.iterations,
               state=CONV_NAMES[r.state], n_corr=r.n_corr, mse=r.mse, prev_mse=r.prev_mse_out)
    if want_corr:
        out["corr_idx"], out["corr_d2"] = idx, d2
    if want_final:
        out["final"] = fin
    if want_trace:
        out["trace"] = tr[:r.iterations]
    return out


class IcpMatch:
    """ICPMatcher::match() + estimateInfo pieces on the oracle."""

    def __init__(self, ref, target, res=-1.0, multiscale_steps=0, params=None, **kw):
        ref, target = _f32(ref), _f32(target)
        self.p = params or icp_params(**kw)
        self.T = np.zeros((4, 4), np.float64)
        self.r = IcpResult()
        self.h = C.c_void_p()
        self.rc = lib().wmo_icp_match(_p(ref, _fp), len(ref), _p(target, _fp), len(target),
                                      C.byref(self.p), C.c_float(res), multiscale_steps,
                                      _p(self.T, _dp), C.byref(self.r), C.byref(self.h))
        self.ok = self.rc == 0

    def __del__(self):
        if getattr(self, "h", None):
            lib().wmo_match_free(self.h)
            self.h = None

    def counts(self):
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        conv = lib().wmo_match_counts(self.h, C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value, bool(conv)

    def _cloud(self, fn, n):
        ptr = fn(self.h)
        return np.ctypeslib.as_array(ptr, shape=(n * 3,)).reshape(n, 3).copy()

    def clouds(self):
        nr, nt, _, _ = self.counts()
        L = lib()
        ref = self._cloud(L.wmo_match_ref, nr)
        tgt = self._cloud(L.wmo_match_target, nt)
        fin = self._cloud(L.wmo_match_final, nr)
        corr = np.ctypeslib.as_array(L.wmo_match_corr(self.h), shape=(nr,)).copy()
        return ref, tgt, fin, corr

    def lum(self):
        info = np.zeros((6, 6))
        rc = lib().wmo_info_lum(self.h, _p(info, _dp))
        return info, rc

    def lumold(self, max_corr=None):
        info = np.zeros((6, 6))
        rc = lib().wmo_info_lumold(self.h, self.p.max_corr if max_corr is None else max_corr,
                                   _p(info, _dp))
        return info, rc

    def censi(self, lin_covar=2.5e-4, ang_covar=7.78e-9):
        info = np.zeros((6, 6))
        rc = lib().wmo_info_censi(self.h, _p(self.T, _dp), lin_covar, ang_covar, _p(info, _dp))
        return info, rc


def lum_from_pairs(p, q):
    p, q = _f32(p), _f32(q)
    info, mm, mz = np.zeros((6, 6)), np.zeros((6, 6)), np.zeros(6)
    ss = C.c_float()
    rc = lib().wmo_lum_from_pairs(_p(p, _fp), _p(q, _fp), len(p), _p(info, _dp), _p(mm, _dp),
                                  _p(mz, _dp), C.byref(ss))
    return dict(rc=rc, info=info, MM=mm, MZ=mz, ss=ss.value)


def censi_from_pairs(ref_pts, tgt_pts, T, lin_covar=2.5e-4, ang_covar=7.78e-9):
    a, b = _f32(ref_pts), _f32(tgt_pts)
    T = np.ascontiguousarray(T, np.float64)
    info, H, mid = np.zeros((6, 6)), np.zeros((6, 6)), np.zeros((6, 6))
    lib().wmo_censi_from_pairs(_p(a, _fp), _p(b, _fp), len(a), _p(T, _dp), lin_covar, ang_covar,
                               _p(info, _dp), _p(H, _dp), _p(mid, _dp))
    return dict(info=info, d2J_dX2=H, middle=mid)


def svd(A):
    A = np.ascontiguousarray(A, np.float64)
    n = A.shape[0]
    U, S, V = np.zeros((n, n)), np.zeros(n), np.zeros((n, n))
    lib().wmo_svd(n, _p(A, _dp), _p(U, _dp), _p(S, _dp), _p(V, _dp))
    return U, S, V


def sym_eig(A):
    A = np.ascontiguousarray(A, np.float64)
    n = A.shape[0]
    w, v = np.zeros(n), np.zeros((n, n))
    lib().wmo_sym_eig(n, _p(A, _dp), _p(w, _dp), _p(v, _dp))
    return w, v


def inverse(A):
    A = np.ascontiguousarray(A, np.float64)
    n = A.shape[0]
    out = np.zeros((n, n))
    lib().wmo_inverse(n, _p(A, _dp), _p(out, _dp))
    return out


def umeyama(src, dst, float_sums=False):
    src, dst = _f32(src), _f32(dst)
    T = np.zeros((4, 4))
    lib().wmo_umeyama(_p(src, _fp), _p(dst, _fp), len(src), int(float_sums), _p(T, _dp))
    return T


def euler_012(R):
    R = np.ascontiguousarray(R, np.float64)
    e = np.zeros(3)
    lib().wmo_euler_angles_012(_p(R, _dp), _p(e, _dp))
    return e


# ---- GICP / NDT wrappers (present once oracle/gicp.c, oracle/ndt.c are built)
def gicp_params(**kw):
    p = GicpParams()
    lib().wmo_gicp_default_params(C.byref(p))
    for k, v in kw.items():
        assert hasattr(p, k), k
        setattr(p, k, v)
    return p


def gicp_covariances(xyz, k=10, eps=1e-3):
    xyz = _f32(xyz)
    cov = np.zeros((len(xyz), 3, 3))
    rc = lib().wmo_gicp_covariances(_p(xyz, _fp), len(xyz), k, eps, _p(cov, _dp))
    assert rc == 0
    return cov


def gicp_set_summation(mode):
    """0: double-double sums (default, what the HIP path reproduces bit for bit); 1: PCL-literal plain
    doubles in index order (oracle/gicp.c).  Process-wide."""
    lib().wmo_gicp_set_summation(int(mode))


def gicp_align(src, tgt, params=None, **kw):
    src, tgt = _f32(src), _f32(tgt)
    p = params or gicp_params(**kw)
    T = np.zeros((4, 4))
    r = GicpResult()
    rc = lib().wmo_gicp_align(_p(src, _fp), len(src), _p(tgt, _fp), len(tgt), C.byref(p),
                              _p(T, _dp), C.byref(r))
    return dict(rc=rc, T=T, converged=bool(r.converged), iterations=r.iterations,
                n_corr=r.n_corr, inner_total=r.inner_total, f=r.f_final, evaluations=r.evaluations)


def gicp_set_objective(mode):
    """0: PCL's per-pair sums through the float transform (default); 1: the same objective as 74 sufficient
    statistics formed once per outer iteration -- the HIP path's default (oracle/gicp.c).  Process-wide."""
    lib().wmo_gicp_set_objective(int(mode))


def gicp_fdf_statistics(src, tgt, src_idx, tgt_idx, mahal, base, T0, x):
    """(f, g, Q): the statistics objective evaluated once (pairs found under the float transform T0)."""
    src, tgt = _f32(src), _f32(tgt)
    si = np.ascontiguousarray(src_idx, np.int32)
    ti = np.ascontiguousarray(tgt_idx, np.int32)
    M = np.ascontiguousarray(mahal, np.float64)
    base = np.ascontiguousarray(base, np.float64)
    T0 = np.ascontiguousarray(T0, np.float32)
    x = np.ascontiguousarray(x, np.float64)
    g, Q = np.zeros(6), np.zeros(74)
    L = lib()
    L.wmo_gicp_fdf_statistics.restype = C.c_double
    L.wmo_gicp_fdf_statistics.argtypes = [_fp, _fp, _ip, _ip, _dp, C.c_int, _dp, _fp, _dp, _dp, _dp]
    f = L.wmo_gicp_fdf_statistics(_p(src, _fp), _p(tgt, _fp), _p(si, _ip), _p(ti, _ip), _p(M, _dp), len(si),
                                  _p(base, _dp), _p(T0, _fp), _p(x, _dp), _p(g, _dp), _p(Q, _dp))
    return f, g, Q


def gicp_fdf(src, tgt, src_idx, tgt_idx, mahal, base, x):
    src, tgt = _f32(src), _f32(tgt)
    si = np.ascontiguousarray(src_idx, np.int32)
    ti = np.ascontiguousarray(tgt_idx, np.int32)
    M = np.ascontiguousarray(mahal, np.float64)
    base = np.ascontiguousarray(base, np.float64)
    x = np.ascontiguousarray(x, np.float64)
    g = np.zeros(6)
    f = lib().wmo_gicp_fdf(_p(src, _fp), _p(tgt, _fp), _p(si, _ip), _p(ti, _ip), _p(M, _dp),
                           len(si), _p(base, _dp), _p(x, _dp), _p(g, _dp))
    return f, g


def ndt_params(**kw):
    p = NdtParams()
    lib().wmo_ndt_default_params(C.byref(p))
    for k, v in kw.items():
        assert hasattr(p, k), k
        setattr(p, k, v)
    return p


class NdtGrid:
    def __init__(self, tgt, res):
        self.tgt = _f32(tgt)
        self.h = lib().wmo_ndt_grid_build(_p(self.tgt, _fp), len(self.tgt), float(res))

    def __del__(self):
        if getattr(self, "h", None):
            lib().wmo_ndt_grid_free(self.h)
            self.h = None

    def size(self):
        return lib().wmo_ndt_grid_size(self.h)

    def export(self):
        V = self.size()
        ijk = np.zeros((V, 3), np.int32)
        mean = np.zeros((V, 3))
        icov = np.zeros((V, 3, 3))
        cnt = np.zeros(V, np.int32)
        lib().wmo_ndt_grid_export(self.h, _p(ijk, _ip), _p(mean, _dp), _p(icov, _dp),
                                  _p(cnt, _ip))
        return ijk, mean, icov, cnt

    def derivatives(self, src, pose, params):
        src = _f32(src)
        pose = np.ascontiguousarray(pose, np.float64)
        g, H = np.zeros(6), np.zeros((6, 6))
        s = lib().wmo_ndt_derivatives(self.h, _p(src, _fp), len(src), C.byref(params),
                                      _p(pose, _dp), _p(g, _dp), _p(H, _dp))
        return s, g, H


def ndt_align(src, tgt, params=None, **kw):
    src, tgt = _f32(src), _f32(tgt)
    p = params or ndt_params(**kw)
    T = np.zeros((4, 4))
    r = NdtResult()
    rc = lib().wmo_ndt_align(_p(src, _fp), len(src), _p(tgt, _fp), len(tgt), C.byref(p),
                             _p(T, _dp), C.byref(r))
    return dict(rc=rc, T=T, converged=bool(r.converged), iterations=r.iterations,
                n_voxels=r.n_voxels, score=r.score)
