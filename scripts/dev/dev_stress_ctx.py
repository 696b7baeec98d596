"""Developer stress: many short-lived contexts, first-call paths (allocations, pinned scratch,
flag waits) exercised over and over."""
import sys, os, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from libwave_amd import capi, synth
mode = sys.argv[1] if len(sys.argv) > 1 else "match"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300_000
ref, tgt, T_gt = synth.pair(n, seed=42)
for k in range(60):
    ctx = capi.Context(0)
    if mode == "match":
        r = ctx.icp_match(ref, tgt, res=0.1, multiscale_steps=0, max_corr=3.0, max_iter=100, t_eps=1e-8, fit_eps=1e-2)
    elif mode == "align":
        ctx.set_source(ref); ctx.set_target(tgt)
        r = ctx.icp_align(max_corr=3.0, max_iter=30, carry_state=0)
    elif mode == "voxel":
        ctx.voxel_downsample(ref, 0.1); r = {"rc": 0}
    assert r["rc"] == 0
    del ctx
    gc.collect()
print("ok", mode)
