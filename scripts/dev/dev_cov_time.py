import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from libwave_amd import capi, synth
ref, tgt, T_gt = synth.pair(500000, seed=42)
d_ref, d_tgt = torch.from_numpy(ref).cuda(), torch.from_numpy(tgt).cuda()
ctx = capi.Context(0)
for _ in range(3):
    ctx.set_source(d_ref); ctx.set_target(d_tgt); r = ctx.gicp_align()
print("ok", r["iterations"])
