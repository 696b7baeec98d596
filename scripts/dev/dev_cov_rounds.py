"""(run under rocprofv3 --kernel-trace; scripts/dev/dev_cov_rounds.sh reads the trace)  k_gicp_cov by cloud size: is its time proportional to the points, or to the ROUNDS of resident workgroups
(86 registers -> 5 waves per SIMD -> 1280 workgroups of 256 queries resident)?"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from libwave_amd import capi, synth
os.environ["WM_GICP_PROFILE"] = "1"
ctx = capi.Context(0)
for n in (327680, 400000, 500000, 655360, 800000, 983040):
    ref, tgt, T_gt = synth.pair(n, seed=42)
    d_ref, d_tgt = torch.from_numpy(ref).cuda(), torch.from_numpy(tgt).cuda()
    ts = []
    for _ in range(4):
        ctx.set_source(d_ref); ctx.set_target(d_tgt); r = ctx.gicp_align()
    print("n %7d  workgroups %5d  rounds %.2f" % (n, (n + 255) // 256, (n + 255) // 256 / 1280.0), flush=True)
