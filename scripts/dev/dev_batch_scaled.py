"""Throughput of wm_icp_batch_match with the reference's voxel-filtered branches on 55k-point scans."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import torch
from libwave_amd import capi, pcd
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
scan = pcd.load_pcd_xyz(os.path.join(root, "tests/golden/testscan.pcd"))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
rng = np.random.default_rng(1)
base = []
for k in range(4):
    tgt = (scan + np.array([0.2 - 0.05 * k, 0.03 * k, 0], np.float32) + rng.uniform(-0.02, 0.02, scan.shape)).astype(np.float32)
    base.append((scan, tgt))
dev = [(torch.from_numpy(r).cuda(), torch.from_numpy(t).cuda()) for r, t in base]
ctx = capi.Context(0)
for res, steps in ((0.1, 3), (0.1, 0)):
    for label, src in (("host", base), ("device", dev)):
        pairs = [src[k % 4] for k in range(B)]
        ctx.icp_batch_match(pairs, with_info=True, res=res, multiscale_steps=steps, max_corr=3.0, max_iter=100)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            got = ctx.icp_batch_match(pairs, with_info=True, res=res, multiscale_steps=steps, max_corr=3.0, max_iter=100)
            ts.append(time.perf_counter() - t0)
        t = float(np.median(ts))
        print("res=%.2f steps=%d B=%d %s: %.1f ms/batch %.0f pairs/s; resident kernels %.2f ms; last-scale iterations %s n_corr %d" % (
            res, steps, B, label, t * 1e3, B / t, got[0]["align_ms"], [g["iterations"] for g in got[:4]], got[0]["n_corr"]), flush=True)
