"""Latency of ICPMatcher::match() with the reference's DEFAULT parameters (res 0.1, three coarser scales)
on the reference's own scan: wm_icp_match + the three estimators."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import torch
from libwave_amd import capi, pcd
scan = pcd.load_pcd_xyz(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests/golden/testscan.pcd"))
T = np.eye(4); T[0, 3] = 0.2
tgt = (scan + np.array([0.2, 0, 0], np.float32)).astype(np.float32)
ctx = capi.Context(0)
for res, steps in ((0.1, 3), (0.1, 0), (-1.0, 0)):
    for _ in range(3):
        r = ctx.icp_match(scan, tgt, res=res, multiscale_steps=steps)
    ts = []
    for _ in range(10):
        t0 = time.perf_counter()
        r = ctx.icp_match(scan, tgt, res=res, multiscale_steps=steps)
        t1 = time.perf_counter()
        ctx.icp_info(capi.WM_INFO_LUM); ctx.icp_info(capi.WM_INFO_CENSI, T_result=r["T"]); ctx.icp_info(capi.WM_INFO_LUMOLD, max_corr=3.0)
        ts.append((t1 - t0, time.perf_counter() - t1))
    m = np.median(np.array(ts), 0) * 1e3
    print("res=%.2f multiscale=%d: match %.3f ms + estimateInfo %.3f ms (iterations %d, n_corr %d, align_ms %.3f)" % (
        res, steps, m[0], m[1], r["iterations"], r["n_corr"], r["align_ms"]), flush=True)
