"""CPU tests (-m "not gpu") of the product's host side: the C-ABI library loads, exports
every symbol include/*.h declares, its host-only solvers agree with the oracle, and it
fails loudly (no fallback) when no HIP device is present."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from helpers import pose_error, svd_stats_numpy
from libwave_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(wm):
    L = wm.lib()
    names = wm.declared_symbols()
    assert len(names) >= 16
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    assert b"gfx950" in L.wm_version()


def test_library_contains_gfx950_code_object():
    so = os.path.join(ROOT, "libwave_amd", "libwavematch_hip.so")
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "-S", so], capture_output=True,
                         text=True).stdout
    assert ".hip_fatbin" in out
    strs = subprocess.run(["strings", "-n", "6", so], capture_output=True, text=True).stdout
    assert "gfx950" in strs


def test_no_cpu_fallback(wm):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a HIP device is present")
    with pytest.raises(wm.WmError):
        wm.Context(0)


def test_batch_entry_points_refuse_bad_arguments_without_a_device(wm):
    """wm_icp_batch_match / wm_gicp_batch_match / wm_ndt_batch_match check their arguments before they touch a device."""
    L = wm.lib()
    status = (ctypes.c_int * 1)()
    items = (wm.BatchItem * 1)()
    assert L.wm_gicp_batch_match(None, items, 1, 16, wm.WM_MEM_HOST, ctypes.byref(wm.gicp_params()), ctypes.c_float(-1.0), None, None,
                                 status, None) == wm.WM_ERR_ARG
    assert L.wm_ndt_batch_match(None, items, 1, 16, wm.WM_MEM_HOST, ctypes.byref(wm.ndt_params()), None, None, status, None) == wm.WM_ERR_ARG
    assert L.wm_icp_batch_match(None, items, 1, 16, wm.WM_MEM_HOST, ctypes.byref(wm.icp_params()), ctypes.c_float(-1.0), 0, 1, None, None,
                                None, status) == wm.WM_ERR_ARG
    assert wm.WM_GICP_BATCH_MAX_POINTS == 100000 and wm.WM_NDT_BATCH_MAX_POINTS == 200000  # (include/wavematch.h)


def test_strerror_and_default_params(wm):
    L = wm.lib()
    assert L.wm_strerror(0) == b"ok"
    assert b"correspondences" in L.wm_strerror(2)
    p = wm.icp_params()
    assert (p.max_corr, p.max_iter, p.t_eps, p.fit_eps) == (3.0, 100, 1e-8, 1e-2)  # icp.hpp:35-43
    assert p.mode == wm.WM_ICP_SVD and p.carry_state == 1


def test_host_umeyama_matches_oracle(wm, oracle):
    src = synth.scene(4000, seed=9)
    T = synth.make_T((0.4, 0.1, -0.3), (0.05, -0.1, 0.2))
    dst = (src.astype(np.float64) @ T[:3, :3].T + T[:3, 3]).astype(np.float32)
    st = svd_stats_numpy(src, dst, np.zeros(len(src), np.float32))
    rc, Tk = wm.umeyama_from_stats(st)
    assert rc == 0
    dt, ang = pose_error(Tk, oracle.umeyama(src, dst))
    assert dt < 1e-9 and ang < 1e-9
    # reflection guard: a mirrored cloud must still give det(R) = +1
    st_m = svd_stats_numpy(src, dst * np.float32([1, 1, -1]), np.zeros(len(src), np.float32))
    rc, Tm = wm.umeyama_from_stats(st_m)
    assert rc == 0 and abs(np.linalg.det(Tm[:3, :3]) - 1) < 1e-9
    # fewer than 3 correspondences
    st[0] = 2
    assert wm.umeyama_from_stats(st)[0] == wm.WM_TOO_FEW


def test_host_gn6_matches_oracle_step(wm, oracle):
    ref, tgt, _ = synth.pair(3000, seed=12)
    idx, d2 = oracle.KdTree(tgt).nn(ref)
    p = ref.astype(np.float64)
    q = tgt[idx].astype(np.float64)
    r = p - q
    J = np.zeros((len(p), 3, 6))
    J[:, 0, 0] = J[:, 1, 1] = J[:, 2, 2] = 1
    J[:, 0, 4], J[:, 0, 5] = p[:, 2], -p[:, 1]
    J[:, 1, 3], J[:, 1, 5] = -p[:, 2], p[:, 0]
    J[:, 2, 3], J[:, 2, 4] = p[:, 1], -p[:, 0]
    H = np.einsum("nca,ncb->ab", J, J)
    g = np.einsum("nca,nc->a", J, r)
    st = np.zeros(32)
    st[0] = len(p)
    st[1] = d2.sum()
    st[2:23] = H[np.triu_indices(6)]
    st[23:29] = g
    rc, Tk = wm.gn6_from_stats(st)
    assert rc == 0
    one = oracle.icp_align(ref, tgt, max_corr=1e3, force_iterations=1, mode=1, incremental_float=0)
    dt, ang = pose_error(Tk, one["T"])
    assert dt < 1e-9 and ang < 1e-9
