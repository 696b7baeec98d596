"""One configuration only (default parameters, device-resident clouds): for kernel-trace profiling."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import torch
from libwave_amd import capi, pcd
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
scan = pcd.load_pcd_xyz(os.path.join(root, "tests/golden/testscan.pcd"))
B = 128
rng = np.random.default_rng(1)
base = []
for k in range(4):
    tgt = (scan + np.array([0.2 - 0.05 * k, 0.03 * k, 0], np.float32) + rng.uniform(-0.02, 0.02, scan.shape)).astype(np.float32)
    base.append((scan, tgt))
dev = [(torch.from_numpy(r).cuda(), torch.from_numpy(t).cuda()) for r, t in base]
ctx = capi.Context(0)
pairs = [dev[k % 4] for k in range(B)]
for _ in range(2):
    ctx.icp_batch_match(pairs, with_info=True, res=0.1, multiscale_steps=3, max_corr=3.0, max_iter=100)
t0 = time.perf_counter()
for _ in range(5):
    ctx.icp_batch_match(pairs, with_info=True, res=0.1, multiscale_steps=3, max_corr=3.0, max_iter=100)
print("ms per call", (time.perf_counter() - t0) / 5 * 1e3)
