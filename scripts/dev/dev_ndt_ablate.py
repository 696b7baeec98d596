"""Where a derivative pass of the NDT 2M ring pair spends its time: the kernel with parts switched off (results wrong,
times informative).  Needs the WM_NDT_ABLATE hook (a developer patch, not in the tree)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from libwave_amd import capi, synth
ref, tgt, T_gt = synth.pair(2_000_000, seed=42, pattern="rings")
d_ref, d_tgt = torch.from_numpy(ref).cuda(), torch.from_numpy(tgt).cuda()
os.environ["WM_TUNE_NDT_GROUP"] = "0"
os.environ["WM_NDT_PROFILE"] = "1"
c = capi.Context(0)
c.set_source(d_ref); c.set_target(d_tgt)
pose = np.zeros(6)
for ab, what in ((0, "whole pass"), (1, "phase 1 only (lists forced empty)"), (2, "no per-point algebra"), (4, "lists forced to 3 entries"), (6, "3 entries, no algebra")):
    os.environ["WM_NDT_ABLATE"] = str(ab)
    for hess in (True, False):
        ts = []
        for _ in range(6):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            c.ndt_derivatives(pose, res=0.5) if hess else c.ndt_derivatives(pose, res=0.5)
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e6)
        print("ablate %d %-36s: %.1f us per evaluation (host wall, incl. fetch)" % (ab, what, np.median(ts)), flush=True)
        break
