"""ctypes binding of libwavematch_hip.so (the C ABI declared in include/wavematch.h).

This is plumbing for tests / bench.py; the product is the shared library.  The
library is built in-tree by `__graft_entry__.build()` (hipcc, gfx950).  There is
no CPU fallback: if the library is missing, or no HIP device is present when a
context is created, the call fails loudly.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libwavematch_hip.so")

WM_OK, WM_NOT_CONVERGED, WM_TOO_FEW = 0, 1, 2
WM_ERR_ARG, WM_ERR_HIP, WM_ERR_RCCL, WM_ERR_STATE, WM_ERR_NOMEM = -1, -2, -3, -4, -5
WM_NDT_BATCH_MAX_POINTS = 200000  # wm_ndt_batch_match: a cloud of a batched NDT registration at most
WM_GICP_BATCH_MAX_POINTS = 100000  # wm_gicp_batch_match: a cloud of a batched GICP registration at most (after the voxel filter)
WM_BATCH_LDS_TARGET_POINTS = 10000  # wm_icp_batch_match: targets up to this size live in one CU's LDS,
WM_BATCH_MAX_TARGET_POINTS = 65535  # larger ones (up to this) in HBM scratch
WM_MEM_HOST, WM_MEM_DEVICE = 0, 1
WM_ICP_SVD, WM_ICP_GN6 = 0, 1
WM_NN_AUTO, WM_NN_GRID, WM_NN_BRUTE = 0, 1, 2
WM_NN_WARM = 0x100
WM_INFO_LUM, WM_INFO_CENSI, WM_INFO_LUMOLD = 0, 1, 2
WM_STATS_LEN = 32
CONV_NAMES = {0: "NOT_CONVERGED", 1: "ITERATIONS", 2: "TRANSFORM", 3: "ABS_MSE", 4: "REL_MSE",
              5: "NO_CORRESPONDENCES", 6: "FORCED"}


class WmError(RuntimeError):
    pass


class IcpParams(C.Structure):
    _fields_ = [("max_corr", C.c_double), ("max_iter", C.c_int), ("t_eps", C.c_double),
                ("fit_eps", C.c_double), ("force_iterations", C.c_int), ("mode", C.c_int),
                ("nn_method", C.c_int), ("carry_state", C.c_int), ("profile", C.c_int),
                ("reserved", C.c_int)]


class IcpStats(C.Structure):
    _fields_ = [("converged", C.c_int), ("iterations", C.c_int), ("state", C.c_int),
                ("n_corr", C.c_int), ("mse", C.c_double), ("prev_mse", C.c_double),
                ("align_ms", C.c_float), ("nn_ms", C.c_float), ("coarse_ms", C.c_float),
                ("stats_ms", C.c_float),
                ("solve_ms", C.c_float), ("nn_launches", C.c_int), ("nn_levels", C.c_int),
                ("deferred", C.c_uint64), ("grid_cell", C.c_float),
                ("owned_violations", C.c_int), ("cert_launches", C.c_int), ("nn_cert_ms", C.c_float),
                ("plan_ms", C.c_float), ("compact_ms", C.c_float), ("index_ms", C.c_float), ("iter_ms", C.c_float),
                ("allreduce_ms", C.c_float), ("n_tgt_local", C.c_uint), ("n_src_local", C.c_uint),
                ("rccl_ranks", C.c_int), ("shard_attempts", C.c_int),
                ("late_iterations", C.c_int), ("late_launches", C.c_int), ("late_ms", C.c_float),
                ("exchange_in_kernel", C.c_int)]


class BatchItem(C.Structure):
    _fields_ = [("src", C.c_void_p), ("n_src", C.c_size_t), ("target", C.c_void_p), ("n_target", C.c_size_t)]


class GicpParams(C.Structure):
    _fields_ = [("corr_rand", C.c_int), ("max_iter", C.c_int), ("r_eps", C.c_double),
                ("t_eps", C.c_double), ("max_corr", C.c_double), ("gicp_epsilon", C.c_double),
                ("max_inner", C.c_int), ("force_iterations", C.c_int), ("objective", C.c_int), ("reserved", C.c_int)]


WM_GICP_OBJECTIVE_PCL_SUMS = 0     # the default (the reference's algorithm): PCL's per-pair sums through the float transform
WM_GICP_OBJECTIVE_STATISTICS = 1   # opt-in: 74 sufficient statistics per outer iteration (csrc/wm_gicp_quad.hpp)


class GicpStats(C.Structure):
    _fields_ = [("converged", C.c_int), ("iterations", C.c_int), ("n_corr", C.c_int),
                ("inner_total", C.c_int), ("evaluations", C.c_int), ("f_final", C.c_double),
                ("fdf_kernel_ms", C.c_float), ("served_evaluations", C.c_int)]


class NdtParams(C.Structure):
    _fields_ = [("res", C.c_double), ("step_size", C.c_double), ("t_eps", C.c_double),
                ("max_iter", C.c_int), ("outlier_ratio", C.c_double),
                ("skip_line_search", C.c_int), ("pcl_d1_sign", C.c_int),
                ("force_iterations", C.c_int)]


class NdtStats(C.Structure):
    _fields_ = [("converged", C.c_int), ("iterations", C.c_int), ("n_voxels", C.c_int),
                ("evaluations", C.c_int), ("score", C.c_double), ("deriv_kernel_ms", C.c_float),
                ("model_builds", C.c_int)]


_dp = C.POINTER(C.c_double)
_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int32)
# int reduce(double *vals, int n, void *user): in-place sum over the ranks (include/wavematch.h)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, _dp, C.c_int, C.c_void_p)
_LIB = None


def lib():
    """Load the shared library (raises if it has not been built)."""
    global _LIB
    if _LIB is None:
        # ONE HIP runtime per process: torch brings its own libamdhip64 and loads it by path -- behind this library's
        # (/opt/rocm's) the process would hold two runtimes that do not know each other's devices, streams or
        # allocations (wm_ctx_create then fails).  With torch loaded first this library binds to torch's copy.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        if not os.path.exists(LIB_PATH):
            raise WmError("libwavematch_hip.so is not built: run `python -c 'import "
                          "__graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950)")
        L = C.CDLL(LIB_PATH)
        L.wm_version.restype = C.c_char_p
        L.wm_strerror.restype = C.c_char_p
        L.wm_strerror.argtypes = [C.c_int]
        L.wm_last_error.restype = C.c_char_p
        L.wm_last_error.argtypes = [C.c_void_p]
        L.wm_ctx_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
        L.wm_ctx_destroy.argtypes = [C.c_void_p]
        L.wm_ctx_destroy.restype = None
        L.wm_set_source.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int]
        L.wm_set_target.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int]
        L.wm_set_grid_cell.argtypes = [C.c_void_p, C.c_float]
        L.wm_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_double]
        L.wm_debug_copy_bandwidth.argtypes = [C.c_void_p, C.c_size_t, C.c_int, _dp]
        L.wm_debug_cert_log.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint), C.c_int]
        L.wm_debug_cert_prof.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
        L.wm_debug_pub_log.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
        L.wm_cloud_sizes.argtypes = [C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        L.wm_icp_default_params.argtypes = [C.POINTER(IcpParams)]
        L.wm_icp_default_params.restype = None
        L.wm_icp_align.argtypes = [C.c_void_p, C.POINTER(IcpParams), _dp, C.POINTER(IcpStats)]
        L.wm_icp_match.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                   C.c_size_t, C.c_int, C.POINTER(IcpParams), C.c_float, C.c_int,
                                   _dp, C.POINTER(IcpStats)]
        L.wm_icp_batch_match.argtypes = [C.c_void_p, C.POINTER(BatchItem), C.c_int, C.c_size_t, C.c_int,
                                         C.POINTER(IcpParams), C.c_float, C.c_int, C.c_int, _dp, _dp,
                                         C.POINTER(IcpStats), C.POINTER(C.c_int)]
        L.wm_gicp_batch_match.argtypes = [C.c_void_p, C.POINTER(BatchItem), C.c_int, C.c_size_t, C.c_int,
                                          C.POINTER(GicpParams), C.c_float, _dp, C.POINTER(GicpStats),
                                          C.POINTER(C.c_int), C.POINTER(C.c_float)]
        L.wm_ndt_batch_match.argtypes = [C.c_void_p, C.POINTER(BatchItem), C.c_int, C.c_size_t, C.c_int,
                                         C.POINTER(NdtParams), _dp, C.POINTER(NdtStats), C.POINTER(C.c_int),
                                         C.POINTER(C.c_float)]
        L.wm_voxel_downsample_batch.argtypes = [C.c_void_p, C.POINTER(BatchItem), C.c_int, C.c_size_t, C.c_int,
                                                C.c_float, _fp, C.c_size_t, C.POINTER(C.c_size_t)]
        L.wm_voxel_downsample.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int,
                                          C.c_float, C.c_void_p, C.c_size_t, C.c_int, C.c_size_t,
                                          C.POINTER(C.c_size_t)]
        L.wm_transform_cloud.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int,
                                         _dp, C.c_void_p, C.c_size_t, C.c_int]
        L.wm_icp_info.argtypes = [C.c_void_p, C.c_int, _dp, C.c_double, C.c_double, C.c_double, _dp,
                                  C.POINTER(C.c_int)]
        L.wm_ctx_set_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.wm_icp_shard_begin.argtypes = [C.c_void_p, C.POINTER(IcpParams), C.c_double, C.c_double,
                                         C.c_size_t]
        L.wm_icp_shard_local_stats.argtypes = [C.c_void_p, C.c_void_p]
        L.wm_icp_shard_apply.argtypes = [C.c_void_p, C.c_void_p]
        L.wm_icp_shard_poll.argtypes = [C.c_void_p, C.POINTER(C.c_int), _dp, C.POINTER(IcpStats)]
        L.wm_host_icp_create.argtypes = [C.POINTER(C.c_void_p), C.POINTER(IcpParams), C.c_size_t]
        L.wm_host_icp_destroy.argtypes = [C.c_void_p]
        L.wm_host_icp_destroy.restype = None
        L.wm_host_icp_apply.argtypes = [C.c_void_p, _dp]
        L.wm_host_icp_get.argtypes = [C.c_void_p, C.POINTER(C.c_int), _dp, C.POINTER(IcpStats)]
        L.wm_gicp_default_params.argtypes = [C.POINTER(GicpParams)]
        L.wm_gicp_default_params.restype = None
        L.wm_gicp_align.argtypes = [C.c_void_p, C.POINTER(GicpParams), _dp, C.POINTER(GicpStats)]
        L.wm_gicp_match.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                    C.c_size_t, C.c_int, C.POINTER(GicpParams), C.c_float, _dp,
                                    C.POINTER(GicpStats)]
        L.wm_gicp_eval.argtypes = [C.c_void_p, C.POINTER(GicpParams), _dp, _dp, _dp, _dp,
                                   C.POINTER(C.c_int)]
        L.wm_gicp_covariances.argtypes = [C.c_void_p, C.c_int, C.c_double, _dp, _dp]
        L.wm_ndt_default_params.argtypes = [C.POINTER(NdtParams)]
        L.wm_ndt_default_params.restype = None
        L.wm_ndt_align.argtypes = [C.c_void_p, C.POINTER(NdtParams), _dp, C.POINTER(NdtStats)]
        L.wm_ndt_derivatives.argtypes = [C.c_void_p, C.POINTER(NdtParams), _dp, _dp, _dp, _dp,
                                         C.POINTER(C.c_int)]
        L.wm_ndt_set_shard.argtypes = [C.c_void_p, C.c_int, C.c_int, ALLREDUCE_FN, C.c_void_p]
        L.wm_get_iteration_times.argtypes = [C.c_void_p, _fp, C.c_int]
        L.wm_comm_get_unique_id.argtypes = [C.c_void_p]
        L.wm_comm_init_rank.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_int, C.c_int]
        L.wm_comm_init_all.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int]
        L.wm_comm_init_local.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int]
        L.wm_comm_destroy.argtypes = [C.c_void_p]
        L.wm_comm_destroy.restype = None
        L.wm_comm_rank.argtypes = [C.c_void_p]
        L.wm_comm_allreduce_probe.argtypes = [C.c_void_p, C.c_void_p, C.c_int, _dp]
        L.wm_comm_world.argtypes = [C.c_void_p]
        L.wm_comm_mailboxes.argtypes = [C.c_void_p]
        L.wm_comm_set_exchange_timeout_ms.argtypes = [C.c_void_p, C.c_int]
        L.wm_icp_align_sharded.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                           C.c_size_t, C.c_size_t, C.c_int, C.POINTER(IcpParams), _dp,
                                           C.POINTER(IcpStats)]
        L.wm_ndt_set_comm.argtypes = [C.c_void_p, C.c_void_p]
        L.wm_ndt_build_model.argtypes = [C.c_void_p, C.c_double]
        L.wm_set_source_filtered.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_float]
        L.wm_set_target_filtered.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_float]
        L.wm_multi_create.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int, C.c_int]
        L.wm_multi_destroy.argtypes = [C.c_void_p]
        L.wm_multi_destroy.restype = None
        L.wm_multi_size.argtypes = [C.c_void_p]
        L.wm_multi_icp_align.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                         C.c_size_t, C.POINTER(IcpParams), _dp, C.POINTER(IcpStats)]
        L.wm_multi_icp_match.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                         C.c_size_t, C.POINTER(IcpParams), C.c_float, C.c_int, _dp, C.POINTER(IcpStats)]
        L.wm_multi_icp_info.argtypes = [C.c_void_p, C.c_int, _dp, C.c_double, C.c_double, C.c_double, _dp,
                                        C.POINTER(C.c_int)]
        L.wm_debug_solve_cycles.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.wm_debug_sort_pairs.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_uint, C.c_void_p]
        L.wm_debug_cost_log.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
        L.wm_debug_phase_log.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.wm_get_correspondences.argtypes = [C.c_void_p, _ip, _fp, C.c_size_t]
        L.wm_nn_search.argtypes = [C.c_void_p, _dp, C.c_double, C.c_int, _ip, _fp, C.c_size_t, _fp]
        L.wm_icp_stats_for.argtypes = [C.c_void_p, _dp, C.c_int, _dp]
        L.wm_umeyama_from_stats.argtypes = [_dp, _dp]
        L.wm_gn6_from_stats.argtypes = [_dp, _dp]
        _LIB = L
    return _LIB


def declared_symbols(header=None):
    """Every function name declared in include/*.h (used by the export test)."""
    import re
    inc = os.path.join(os.path.dirname(_HERE), "include")
    names = []
    for fn in sorted(os.listdir(inc)):
        if not fn.endswith(".h"):
            continue
        txt = open(os.path.join(inc, fn)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        names += re.findall(r"\b(wm_[a-z0-9_]+)\s*\(", txt)
    return sorted(set(names))


def gicp_params(**kw):
    p = GicpParams()
    lib().wm_gicp_default_params(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def ndt_params(**kw):
    p = NdtParams()
    lib().wm_ndt_default_params(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def icp_params(**kw):
    p = IcpParams()
    lib().wm_icp_default_params(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def _cloud_arg(a):
    """-> (pointer, n, stride, mem, keepalive).  Accepts a float32 numpy array
    (n, 3|4) on the host or a torch tensor (n, 3|4) float32 on a HIP device."""
    if isinstance(a, np.ndarray):
        a = np.ascontiguousarray(a, dtype=np.float32)
        assert a.ndim == 2 and a.shape[1] in (3, 4)
        return a.ctypes.data, a.shape[0], a.shape[1] * 4, WM_MEM_HOST, a
    import torch
    assert isinstance(a, torch.Tensor) and a.dtype == torch.float32 and a.dim() == 2
    a = a.contiguous()
    mem = WM_MEM_DEVICE if a.is_cuda else WM_MEM_HOST
    return a.data_ptr(), a.shape[0], a.shape[1] * 4, mem, a


class Context:
    """One wm_ctx == one reference matcher object."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        rc = lib().wm_ctx_create(C.byref(self._h), int(device))
        if rc != WM_OK:
            raise WmError("wm_ctx_create(device=%d) failed: %s -- a HIP device is required; "
                          "there is no CPU fallback" % (device, lib().wm_strerror(rc).decode()))
        self.device = device
        self.n_src = self.n_tgt = 0

    def close(self):
        if self._h:
            lib().wm_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc < 0:
            raise WmError("%s: %s [%s]" % (what, lib().wm_strerror(rc).decode(),
                                           lib().wm_last_error(self._h).decode()))
        return rc

    def set_source(self, cloud):
        ptr, n, stride, mem, keep = _cloud_arg(cloud)
        self._check(lib().wm_set_source(self._h, C.c_void_p(ptr), n, stride, mem), "wm_set_source")
        self.n_src = n

    def set_target(self, cloud):
        ptr, n, stride, mem, keep = _cloud_arg(cloud)
        self._check(lib().wm_set_target(self._h, C.c_void_p(ptr), n, stride, mem), "wm_set_target")
        self.n_tgt = n

    def copy_bandwidth(self, nbytes=1 << 30, reps=10):
        """GB/s (read + write) of a float4 device-to-device copy kernel on this GPU."""
        out = C.c_double(0)
        self._check(lib().wm_debug_copy_bandwidth(self._h, int(nbytes), int(reps), C.byref(out)),
                    "wm_debug_copy_bandwidth")
        return out.value

    def set_option(self, name, value):
        self._check(lib().wm_set_option(self._h, name.encode(), C.c_double(value)), "wm_set_option")

    def cert_log(self, iterations=None):
        """cert_log(n): arm a log of n launches of the certificate kernel; cert_log(): the number of
        queries each launch since had to search."""
        if iterations is not None:
            self._check(lib().wm_debug_cert_log(self._h, int(iterations), None, 0), "wm_debug_cert_log")
            return None
        out = (C.c_uint * 4096)()
        k = lib().wm_debug_cert_log(self._h, 0, out, 4096)
        return [int(out[i]) for i in range(max(k, 0))]

    def pub_log(self):
        """[(iteration, step size m, fraction of matches changed, fraction searched by the certificate kernel)]"""
        out = (C.c_uint64 * 1024)()
        n = lib().wm_debug_pub_log(self._h, out, 1024)
        rows = []
        for k in range(1, max(n, 0)):
            w = int(out[k])
            if w == 0:
                break
            disp = np.array([((w >> 32) & 0xFFFF) << 16], np.uint32).view(np.float32)[0]
            rows.append((w >> 48, float(disp), ((w >> 16) & 0xFFFF) / 65535.0, (w & 0xFFFF) / 65535.0))
        return rows

    def cert_prof(self):
        out = (C.c_uint64 * (64 * 128))()
        k = lib().wm_debug_cert_prof(self._h, out, 128)
        return np.array(out[:64 * max(k, 0)], dtype=np.float64).reshape(-1, 4, 16)  # [launch][sampled workgroup][stamp]

    def set_grid_cell(self, h):
        self._check(lib().wm_set_grid_cell(self._h, float(h)), "wm_set_grid_cell")

    def sizes(self):
        a, b = C.c_size_t(), C.c_size_t()
        lib().wm_cloud_sizes(self._h, C.byref(a), C.byref(b))
        return a.value, b.value

    def icp_align(self, params=None, **kw):
        p = params or icp_params(**kw)
        T = np.zeros((4, 4), np.float64)
        s = IcpStats()
        rc = self._check(lib().wm_icp_align(self._h, C.byref(p), T.ctypes.data_as(_dp),
                                            C.byref(s)), "wm_icp_align")
        return self._stats_dict(rc, T, s)

    @staticmethod
    def _stats_dict(rc, T, s):
        return dict(rc=rc, T=T if rc == WM_OK else None, converged=bool(s.converged),
                    iterations=s.iterations, state=CONV_NAMES.get(s.state, s.state),
                    n_corr=s.n_corr, mse=s.mse, prev_mse=s.prev_mse, align_ms=s.align_ms,
                    nn_ms=s.nn_ms, coarse_ms=s.coarse_ms, stats_ms=s.stats_ms,
                    solve_ms=s.solve_ms, nn_launches=s.nn_launches, nn_levels=s.nn_levels,
                    deferred=s.deferred, grid_cell=s.grid_cell,
                    owned_violations=s.owned_violations, cert_launches=s.cert_launches,
                    nn_cert_ms=s.nn_cert_ms, late_iterations=s.late_iterations, late_launches=s.late_launches,
                    late_ms=s.late_ms)

    def icp_match(self, ref, target, res=-1.0, multiscale_steps=0, params=None, **kw):
        """ICPMatcher::match() (icp.cpp:75-133) in one C-ABI call."""
        p = params or icp_params(**kw)
        pr, nr, sr, mr, k1 = _cloud_arg(ref)
        pt, nt, stt, mt, k2 = _cloud_arg(target)
        assert sr == stt and mr == mt
        T = np.zeros((4, 4), np.float64)
        s = IcpStats()
        rc = self._check(lib().wm_icp_match(self._h, C.c_void_p(pr), nr, C.c_void_p(pt), nt, sr, mr,
                                            C.byref(p), C.c_float(res), int(multiscale_steps),
                                            T.ctypes.data_as(_dp), C.byref(s)), "wm_icp_match")
        self.n_src, self.n_tgt = self.sizes()
        return self._stats_dict(rc, T, s)

    def icp_batch_match(self, pairs, with_info=True, res=-1.0, multiscale_steps=0, params=None, **kw):
        """Many registrations (+ estimateLUMold) per launch (wm_icp_batch_match): ICPMatcher::match()
        of every pair, res / multiscale_steps as icp_match's.  pairs: [(ref, target), ...]; -> list of
        dicts as icp_align's, plus 'info' (6x6) when with_info."""
        p = params or icp_params(**kw)
        n = len(pairs)
        items = (BatchItem * max(n, 1))()
        keep = []
        stride = mem = None
        for k, (ref, tgt) in enumerate(pairs):
            pr, nr, sr, mr, k1 = _cloud_arg(ref)
            pt, nt, stt, mt, k2 = _cloud_arg(tgt)
            assert sr == stt and mr == mt and (stride in (None, sr)) and (mem in (None, mr))
            stride, mem = sr, mr
            keep += [k1, k2]
            items[k].src, items[k].n_src, items[k].target, items[k].n_target = pr, nr, pt, nt
        T = np.zeros((max(n, 1), 4, 4), np.float64)
        info = np.zeros((max(n, 1), 6, 6), np.float64)
        stats = (IcpStats * max(n, 1))()
        status = (C.c_int * max(n, 1))()
        self._check(lib().wm_icp_batch_match(self._h, items, n, stride or 16, mem or WM_MEM_HOST, C.byref(p),
                                             C.c_float(res), int(multiscale_steps),
                                             1 if with_info else 0, T.ctypes.data_as(_dp),
                                             info.ctypes.data_as(_dp), stats, status), "wm_icp_batch_match")
        out = []
        for k in range(n):
            d = self._stats_dict(status[k], T[k].copy(), stats[k])
            if with_info:
                d["info"] = info[k].copy()
            out.append(d)
        return out

    def voxel_downsample_batch(self, pairs, leaf):
        """pcl::VoxelGrid of all clouds of `pairs` in one pass -> [(filtered ref, filtered target), ...]."""
        n = len(pairs)
        items = (BatchItem * max(n, 1))()
        keep, stride, mem, cap = [], None, None, 0
        for k, (ref, tgt) in enumerate(pairs):
            pr, nr, sr, mr, k1 = _cloud_arg(ref)
            pt, nt, stt, mt, k2 = _cloud_arg(tgt)
            assert sr == stt and mr == mt and (stride in (None, sr)) and (mem in (None, mr))
            stride, mem = sr, mr
            keep += [k1, k2]
            items[k].src, items[k].n_src, items[k].target, items[k].n_target = pr, nr, pt, nt
            cap += nr + nt
        out = np.empty((max(cap, 1), 3), np.float32)
        counts = (C.c_size_t * max(2 * n, 1))()
        self._check(lib().wm_voxel_downsample_batch(self._h, items, n, stride or 16, mem or WM_MEM_HOST, C.c_float(leaf),
                                                    out.ctypes.data_as(_fp), cap, counts), "wm_voxel_downsample_batch")
        res, w = [], 0
        for k in range(n):
            a = out[w:w + counts[2 * k]].copy()
            w += counts[2 * k]
            b = out[w:w + counts[2 * k + 1]].copy()
            w += counts[2 * k + 1]
            res.append((a, b))
        return res

    def voxel_downsample(self, cloud, leaf):
        ptr, n, stride, mem, keep = _cloud_arg(cloud)
        out = np.empty((max(n, 1), 3), np.float32)
        m = C.c_size_t(0)
        self._check(lib().wm_voxel_downsample(self._h, C.c_void_p(ptr), n, stride, mem,
                                              C.c_float(leaf), C.c_void_p(out.ctypes.data), 12,
                                              WM_MEM_HOST, len(out), C.byref(m)),
                    "wm_voxel_downsample")
        return out[:m.value].copy()

    def transform_cloud(self, cloud, T):
        ptr, n, stride, mem, keep = _cloud_arg(cloud)
        T = np.ascontiguousarray(T, np.float64)
        out = np.empty((n, 3), np.float32)
        self._check(lib().wm_transform_cloud(self._h, C.c_void_p(ptr), n, stride, mem,
                                             T.ctypes.data_as(_dp), C.c_void_p(out.ctypes.data), 12,
                                             WM_MEM_HOST), "wm_transform_cloud")
        return out

    def icp_info(self, method, T_result=None, lin_covar=2.5e-4, ang_covar=7.78e-9, max_corr=3.0):
        """estimateLUM / estimateCensi / estimateLUMold on the last align's correspondences."""
        info = np.zeros((6, 6), np.float64)
        deg = C.c_int(0)
        T = None if T_result is None else np.ascontiguousarray(T_result, np.float64)
        rc = self._check(lib().wm_icp_info(self._h, int(method),
                                           T.ctypes.data_as(_dp) if T is not None else None,
                                           float(lin_covar), float(ang_covar), float(max_corr),
                                           info.ctypes.data_as(_dp), C.byref(deg)), "wm_icp_info")
        return rc, info, bool(deg.value)

    # ---- GICP
    @staticmethod
    def _gicp_dict(rc, T, s):
        return dict(rc=rc, T=T if rc == WM_OK else None, converged=bool(s.converged),
                    iterations=s.iterations, n_corr=s.n_corr, inner_total=s.inner_total,
                    evaluations=s.evaluations, f=s.f_final, fdf_kernel_ms=s.fdf_kernel_ms,
                    served_evaluations=s.served_evaluations)

    def gicp_align(self, params=None, **kw):
        p = params or gicp_params(**kw)
        T = np.zeros((4, 4), np.float64)
        s = GicpStats()
        rc = self._check(lib().wm_gicp_align(self._h, C.byref(p), T.ctypes.data_as(_dp),
                                             C.byref(s)), "wm_gicp_align")
        return self._gicp_dict(rc, T, s)

    def gicp_match(self, ref, target, res=-1.0, params=None, **kw):
        p = params or gicp_params(**kw)
        pr, nr, sr, mr, k1 = _cloud_arg(ref)
        pt, nt, stt, mt, k2 = _cloud_arg(target)
        assert sr == stt and mr == mt
        T = np.zeros((4, 4), np.float64)
        s = GicpStats()
        rc = self._check(lib().wm_gicp_match(self._h, C.c_void_p(pr), nr, C.c_void_p(pt), nt, sr, mr,
                                             C.byref(p), C.c_float(res), T.ctypes.data_as(_dp),
                                             C.byref(s)), "wm_gicp_match")
        self.n_src, self.n_tgt = self.sizes()
        return self._gicp_dict(rc, T, s)

    def gicp_batch_match(self, pairs, res=-1.0, params=None, **kw):
        """GICPMatcher::match() of every pair in one launch (wm_gicp_batch_match), one registration per compute
        unit.  pairs: [(ref, target), ...] -> list of dicts as gicp_align's (+ 'kernel_ms' of the launch)."""
        p = params or gicp_params(**kw)
        n = len(pairs)
        items = (BatchItem * max(n, 1))()
        keep = []
        stride = mem = None
        for k, (ref, tgt) in enumerate(pairs):
            pr, nr, sr, mr, k1 = _cloud_arg(ref)
            pt, nt, stt, mt, k2 = _cloud_arg(tgt)
            assert sr == stt and mr == mt and (stride in (None, sr)) and (mem in (None, mr))
            stride, mem = sr, mr
            keep += [k1, k2]
            items[k].src, items[k].n_src, items[k].target, items[k].n_target = pr, nr, pt, nt
        T = np.zeros((max(n, 1), 4, 4), np.float64)
        stats = (GicpStats * max(n, 1))()
        status = (C.c_int * max(n, 1))()
        ms = C.c_float(0)
        self._check(lib().wm_gicp_batch_match(self._h, items, n, stride or 16, mem or WM_MEM_HOST, C.byref(p),
                                              C.c_float(res), T.ctypes.data_as(_dp), stats, status, C.byref(ms)),
                    "wm_gicp_batch_match")
        out = []
        for k in range(n):
            d = self._gicp_dict(status[k], T[k].copy(), stats[k])
            d["kernel_ms"] = ms.value
            out.append(d)
        return out

    def gicp_eval(self, T_pair, x, params=None, **kw):
        p = params or gicp_params(**kw)
        T = np.ascontiguousarray(T_pair, np.float64)
        x = np.ascontiguousarray(x, np.float64)
        f = C.c_double(0)
        g = np.zeros(6)
        m = C.c_int(0)
        self._check(lib().wm_gicp_eval(self._h, C.byref(p), T.ctypes.data_as(_dp),
                                       x.ctypes.data_as(_dp), C.byref(f), g.ctypes.data_as(_dp),
                                       C.byref(m)), "wm_gicp_eval")
        return f.value, g, m.value

    def gicp_covariances(self, k=10, eps=1e-3):
        cs = np.zeros((self.n_src, 3, 3), np.float64)
        ct = np.zeros((self.n_tgt, 3, 3), np.float64)
        self._check(lib().wm_gicp_covariances(self._h, int(k), float(eps), cs.ctypes.data_as(_dp),
                                              ct.ctypes.data_as(_dp)), "wm_gicp_covariances")
        return cs, ct

    # ---- NDT
    def ndt_align(self, params=None, **kw):
        p = params or ndt_params(**kw)
        T = np.zeros((4, 4), np.float64)
        s = NdtStats()
        rc = self._check(lib().wm_ndt_align(self._h, C.byref(p), T.ctypes.data_as(_dp),
                                            C.byref(s)), "wm_ndt_align")
        return dict(rc=rc, T=T if rc == WM_OK else None, converged=bool(s.converged),
                    iterations=s.iterations, n_voxels=s.n_voxels, evaluations=s.evaluations,
                    score=s.score, deriv_kernel_ms=s.deriv_kernel_ms, model_builds=s.model_builds)

    def ndt_batch_match(self, pairs, params=None, **kw):
        """NDTMatcher::match() of every pair in one launch (wm_ndt_batch_match), one registration per compute unit.
        pairs: [(ref, target), ...] -> list of dicts as ndt_align's (+ 'kernel_ms' of the launch)."""
        p = params or ndt_params(**kw)
        n = len(pairs)
        items = (BatchItem * max(n, 1))()
        keep = []
        stride = mem = None
        for k, (ref, tgt) in enumerate(pairs):
            pr, nr, sr, mr, k1 = _cloud_arg(ref)
            pt, nt, stt, mt, k2 = _cloud_arg(tgt)
            assert sr == stt and mr == mt and (stride in (None, sr)) and (mem in (None, mr))
            stride, mem = sr, mr
            keep += [k1, k2]
            items[k].src, items[k].n_src, items[k].target, items[k].n_target = pr, nr, pt, nt
        T = np.zeros((max(n, 1), 4, 4), np.float64)
        stats = (NdtStats * max(n, 1))()
        status = (C.c_int * max(n, 1))()
        ms = C.c_float(0)
        self._check(lib().wm_ndt_batch_match(self._h, items, n, stride or 16, mem or WM_MEM_HOST, C.byref(p),
                                             T.ctypes.data_as(_dp), stats, status, C.byref(ms)), "wm_ndt_batch_match")
        out = []
        for k in range(n):
            s = stats[k]
            out.append(dict(rc=status[k], T=T[k].copy() if status[k] == WM_OK else None, converged=bool(s.converged),
                            iterations=s.iterations, n_voxels=s.n_voxels, evaluations=s.evaluations, score=s.score,
                            model_builds=s.model_builds, kernel_ms=ms.value))
        return out

    def ndt_derivatives(self, pose, params=None, **kw):
        p = params or ndt_params(**kw)
        pose = np.ascontiguousarray(pose, np.float64)
        score = C.c_double(0)
        g, H = np.zeros(6), np.zeros((6, 6))
        nv = C.c_int(0)
        self._check(lib().wm_ndt_derivatives(self._h, C.byref(p), pose.ctypes.data_as(_dp),
                                             C.byref(score), g.ctypes.data_as(_dp),
                                             H.ctypes.data_as(_dp), C.byref(nv)),
                    "wm_ndt_derivatives")
        return score.value, g, H, nv.value

    def ndt_set_shard(self, rank, world, reduce=None):
        """This context evaluates slice `rank` of `world` of the source in every NDT derivative
        pass; `reduce` = an ALLREDUCE_FN (see sharding.make_allreduce) that sums the 28 pass totals
        over the ranks.  The context keeps the callback alive."""
        self._ndt_reduce = reduce if reduce is not None else ALLREDUCE_FN(0)
        self._check(lib().wm_ndt_set_shard(self._h, int(rank), int(world), self._ndt_reduce, None),
                    "wm_ndt_set_shard")

    # ---- sharded (multi-GPU) stepping
    def set_stream(self, stream_ptr, external=True):
        self._check(lib().wm_ctx_set_stream(self._h, C.c_void_p(stream_ptr), int(external)),
                    "wm_ctx_set_stream")

    def shard_begin(self, params, x_lo, x_hi, expect_owned_total=0):
        self._check(lib().wm_icp_shard_begin(self._h, C.byref(params), float(x_lo), float(x_hi),
                                             int(expect_owned_total)), "wm_icp_shard_begin")

    def shard_local_stats(self, dev_ptr):
        self._check(lib().wm_icp_shard_local_stats(self._h, C.c_void_p(dev_ptr)),
                    "wm_icp_shard_local_stats")

    def shard_apply(self, dev_ptr):
        self._check(lib().wm_icp_shard_apply(self._h, C.c_void_p(dev_ptr)), "wm_icp_shard_apply")

    def shard_poll(self):
        done = C.c_int(0)
        T = np.zeros((4, 4), np.float64)
        s = IcpStats()
        rc = self._check(lib().wm_icp_shard_poll(self._h, C.byref(done), T.ctypes.data_as(_dp),
                                                 C.byref(s)), "wm_icp_shard_poll")
        d = self._stats_dict(rc, T, s)
        d["T"] = T
        d["done"] = bool(done.value)
        return d

    def icp_align_sharded(self, comm, ref, target, params=None, **kw):
        """wm_icp_align_sharded: one registration over all ranks of `comm` (a Comm or None)."""
        p = params or icp_params(**kw)
        rp, rn, rs, rm, k1 = _cloud_arg(ref)
        tp, tn, ts, tm, k2 = _cloud_arg(target)
        if rs != ts or rm != tm:
            raise WmError("ref and target must share stride and memory space")
        T = np.zeros((4, 4), np.float64)
        s = IcpStats()
        rc = self._check(lib().wm_icp_align_sharded(self._h, comm.handle if comm is not None else None,
                                                    C.c_void_p(rp), rn, C.c_void_p(tp), tn, rs, rm,
                                                    C.byref(p), T.ctypes.data_as(_dp), C.byref(s)),
                         "wm_icp_align_sharded")
        return self._sharded_dict(rc, T, s)

    @staticmethod
    def _sharded_dict(rc, T, s):
        """_stats_dict + what a sharded registration says about itself (wm_icp_stats' planning / exchange fields)"""
        d = Context._stats_dict(rc, T, s)
        d["owned_violations"] = s.owned_violations
        for k in ("plan_ms", "compact_ms", "index_ms", "iter_ms", "allreduce_ms", "n_tgt_local", "n_src_local",
                  "rccl_ranks", "shard_attempts", "exchange_in_kernel"):
            d[k] = getattr(s, k)
        return d

    def allreduce_probe(self, comm, reps=50):
        """us per all-reduce of the 32-double block on `comm`, back to back on this context's stream."""
        out = C.c_double(0)
        self._check(lib().wm_comm_allreduce_probe(self._h, comm.handle, int(reps), C.byref(out)),
                    "wm_comm_allreduce_probe")
        return out.value

    def ndt_set_comm(self, comm):
        self._check(lib().wm_ndt_set_comm(self._h, comm.handle if comm is not None else None),
                    "wm_ndt_set_comm")

    def cost_log_arm(self, iterations):
        self._check(lib().wm_debug_cost_log(self._h, iterations, None, 0), "wm_debug_cost_log")

    def cost_log_fetch(self, iterations, n):
        buf = np.zeros((iterations, n), np.uint32)
        k = lib().wm_debug_cost_log(self._h, iterations, buf.ctypes.data_as(C.c_void_p), buf.size)
        if k < 0:
            raise WmError("wm_debug_cost_log: %d" % k)
        return buf[:k]

    def phase_log_fetch(self, iterations):
        buf = np.zeros((iterations, 8), np.uint64)
        k = lib().wm_debug_phase_log(self._h, buf.ctypes.data_as(C.c_void_p), iterations)
        return buf[:max(k, 0)]

    def sort_pairs(self, keys, bits):
        """The library's own radix sort (wm_debug_sort_pairs): positions of `keys` (uint32 / uint64 numpy array) in
        ascending order of their low `bits` bits, equal keys in input order."""
        keys = np.ascontiguousarray(keys)
        assert keys.dtype in (np.uint32, np.uint64) and keys.ndim == 1
        out = np.zeros(keys.shape[0], np.uint32)
        self._check(lib().wm_debug_sort_pairs(self._h, keys.ctypes.data, keys.dtype.itemsize, keys.shape[0], int(bits),
                                              out.ctypes.data), "wm_debug_sort_pairs")
        return out

    def solve_cycles(self):
        buf = (C.c_uint64 * 8)()
        lib().wm_debug_solve_cycles(self._h, buf)
        return [int(v) for v in buf]

    def iteration_times(self, cap=1024):
        buf = np.zeros(cap, np.float32)
        n = lib().wm_get_iteration_times(self._h, buf.ctypes.data_as(_fp), cap)
        return buf[:n].copy()

    def correspondences(self):
        idx = np.empty(self.n_src, np.int32)
        d2 = np.empty(self.n_src, np.float32)
        self._check(lib().wm_get_correspondences(self._h, idx.ctypes.data_as(_ip),
                                                 d2.ctypes.data_as(_fp), self.n_src),
                    "wm_get_correspondences")
        return idx, d2

    def nn_search(self, T=None, max_corr=3.0, nn_method=WM_NN_AUTO, want=True, timed=False):
        T = np.ascontiguousarray(np.eye(4) if T is None else T, np.float64)
        idx = np.empty(self.n_src, np.int32) if want else None
        d2 = np.empty(self.n_src, np.float32) if want else None
        ms = C.c_float(0)
        self._check(lib().wm_nn_search(self._h, T.ctypes.data_as(_dp), float(max_corr),
                                       int(nn_method), idx.ctypes.data_as(_ip) if want else None,
                                       d2.ctypes.data_as(_fp) if want else None, self.n_src,
                                       C.byref(ms) if timed else None), "wm_nn_search")
        return (idx, d2, ms.value) if timed else (idx, d2)

    def icp_stats_for(self, T, mode=WM_ICP_SVD):
        T = np.ascontiguousarray(T, np.float64)
        st = np.zeros(WM_STATS_LEN, np.float64)
        self._check(lib().wm_icp_stats_for(self._h, T.ctypes.data_as(_dp), int(mode),
                                           st.ctypes.data_as(_dp)), "wm_icp_stats_for")
        return st


class HostIcp:
    """The per-iteration solve + PCL stopping rules on the host (wm_host_icp_*): the same
    function the device runs after the all-reduce."""

    def __init__(self, params, expect_owned_total=0):
        self._h = C.c_void_p()
        rc = lib().wm_host_icp_create(C.byref(self._h), C.byref(params), int(expect_owned_total))
        if rc != WM_OK:
            raise WmError("wm_host_icp_create failed")

    def __del__(self):
        if getattr(self, "_h", None):
            lib().wm_host_icp_destroy(self._h)
            self._h = None

    def apply(self, stats):
        st = np.ascontiguousarray(stats, np.float64)
        assert st.size == WM_STATS_LEN
        lib().wm_host_icp_apply(self._h, st.ctypes.data_as(_dp))

    def get(self):
        done = C.c_int(0)
        T = np.zeros((4, 4), np.float64)
        s = IcpStats()
        lib().wm_host_icp_get(self._h, C.byref(done), T.ctypes.data_as(_dp), C.byref(s))
        return dict(done=bool(done.value), T=T, converged=bool(s.converged),
                    iterations=s.iterations, state=CONV_NAMES.get(s.state, s.state),
                    n_corr=s.n_corr, mse=s.mse, owned_violations=s.owned_violations,
                    rc=0 if s.converged or not done.value else
                    (WM_TOO_FEW if s.state == 5 else WM_NOT_CONVERGED))


def umeyama_from_stats(stats):
    st = np.ascontiguousarray(stats, np.float64)
    T = np.zeros((4, 4))
    rc = lib().wm_umeyama_from_stats(st.ctypes.data_as(_dp), T.ctypes.data_as(_dp))
    return rc, T


def gn6_from_stats(stats):
    st = np.ascontiguousarray(stats, np.float64)
    T = np.zeros((4, 4))
    rc = lib().wm_gn6_from_stats(st.ctypes.data_as(_dp), T.ctypes.data_as(_dp))
    return rc, T


class Comm:
    """One rank's wm_comm (RCCL communicator, or the single-GPU stand-in)."""

    def __init__(self, handle):
        self.handle = C.c_void_p(handle)

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        rc = lib().wm_comm_get_unique_id(buf)
        if rc != 0:
            raise WmError("wm_comm_get_unique_id: %d" % rc)
        return buf.raw

    @classmethod
    def init_rank(cls, device, uid, rank, world):
        h = C.c_void_p()
        buf = C.create_string_buffer(bytes(uid), 128)
        rc = lib().wm_comm_init_rank(C.byref(h), device, buf, rank, world)
        if rc != 0:
            raise WmError("wm_comm_init_rank: %d" % rc)
        return cls(h.value)

    @classmethod
    def init_local(cls, n, device=0):
        arr = (C.c_void_p * n)()
        rc = lib().wm_comm_init_local(arr, n, device)
        if rc != 0:
            raise WmError("wm_comm_init_local: %d" % rc)
        return [cls(arr[i]) for i in range(n)]

    @property
    def rank(self):
        return lib().wm_comm_rank(self.handle)

    @property
    def world(self):
        return lib().wm_comm_world(self.handle)

    @property
    def mailboxes(self):
        """True while the sharded loop's exchange goes through the ranks' mailboxes (wm_comm_mailboxes)."""
        return bool(lib().wm_comm_mailboxes(self.handle))

    def set_exchange_timeout_ms(self, ms):
        rc = lib().wm_comm_set_exchange_timeout_ms(self.handle, int(ms))
        if rc != 0:
            raise WmError("wm_comm_set_exchange_timeout_ms: %d" % rc)

    def close(self):
        if self.handle:
            lib().wm_comm_destroy(self.handle)
            self.handle = None


class Multi:
    """wm_multi: all ranks of a sharded registration in this process (one thread per device)."""

    def __init__(self, devices, emulate=False):
        arr = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        rc = lib().wm_multi_create(C.byref(h), arr, len(devices), int(emulate))
        if rc != 0:
            raise WmError("wm_multi_create: %d (%s)" % (rc, lib().wm_strerror(rc).decode()))
        self._h = h

    def icp_align(self, ref, target, params=None, **kw):
        return self.icp_match(ref, target, res=-1.0, multiscale_steps=0, params=params, **kw)

    def icp_info(self, method, T=None, lin_covar=0.0, ang_covar=0.0, max_corr=0.0):
        """wm_multi_icp_info: the estimators after a registration over the group -> (rc, info 6x6, degenerate)"""
        info = np.zeros((6, 6), np.float64)
        deg = C.c_int(0)
        Tp = None if T is None else np.ascontiguousarray(T, np.float64).ctypes.data_as(_dp)
        rc = lib().wm_multi_icp_info(self._h, int(method), Tp, lin_covar, ang_covar, max_corr,
                                     info.ctypes.data_as(_dp), C.byref(deg))
        if rc < 0:
            raise WmError("wm_multi_icp_info: %d (%s)" % (rc, lib().wm_strerror(rc).decode()))
        return rc, info, deg.value

    def icp_match(self, ref, target, res=-1.0, multiscale_steps=0, params=None, **kw):
        p = params or icp_params(**kw)
        ref = np.ascontiguousarray(ref, np.float32)
        target = np.ascontiguousarray(target, np.float32)
        T = np.zeros((4, 4), np.float64)
        s = IcpStats()
        rc = lib().wm_multi_icp_match(self._h, ref.ctypes.data_as(C.c_void_p), len(ref),
                                      target.ctypes.data_as(C.c_void_p), len(target), ref.strides[0],
                                      C.byref(p), C.c_float(res), int(multiscale_steps), T.ctypes.data_as(_dp),
                                      C.byref(s))
        if rc < 0:
            raise WmError("wm_multi_icp_match: %d (%s)" % (rc, lib().wm_strerror(rc).decode()))
        d = dict(rc=rc, T=T, converged=s.converged, iterations=s.iterations, state=s.state,
                 n_corr=s.n_corr, mse=s.mse, align_ms=s.align_ms, owned_violations=s.owned_violations,
                 cert_launches=s.cert_launches)
        for k in ("plan_ms", "compact_ms", "index_ms", "iter_ms", "allreduce_ms", "n_tgt_local", "n_src_local",
                  "rccl_ranks", "shard_attempts", "exchange_in_kernel"):
            d[k] = getattr(s, k)   # (rank 0's)
        return d

    def close(self):
        if self._h:
            lib().wm_multi_destroy(self._h)
            self._h = None
