"""NDT 2M-point 64-ring pair, 0.5 m voxels (BASELINE configs[3]): ms per registration and us per derivative pass by
the number of workgroups of a pass (WM_TUNE_NDT_BLOCKS; 0 = one resident round, the default).
NDT_PATTERN= NDT_POINTS=1000000 NDT_RES=1.0 for the uniform scene."""
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import torch

from libwave_amd import capi, synth

n = int(os.environ.get("NDT_POINTS", "2000000"))
pattern = os.environ.get("NDT_PATTERN", "rings")
ref, tgt, T_gt = synth.pair(n, seed=42, pattern=pattern) if pattern else synth.pair(n, seed=42)
d_ref, d_tgt = torch.from_numpy(ref).cuda(), torch.from_numpy(tgt).cuda()
os.environ["WM_TRACE"] = os.environ.get("WM_TRACE", "0")
for group in [int(x) for x in os.environ.get("NDT_BLOCKS", "0,512,768,1024,1536").split(",")]:
    os.environ["WM_TUNE_NDT_BLOCKS"] = str(group)
    ctx = capi.Context(0)
    os.environ["WM_NDT_PROFILE"] = "1"
    prof = capi.Context(0)
    os.environ.pop("WM_NDT_PROFILE")

    def run(c, **kw):
        c.set_source(d_ref)
        c.set_target(d_tgt)
        return c.ndt_align(res=float(os.environ.get("NDT_RES", "0.5")), **kw)
    run(ctx)
    ts = []
    for _ in range(7):
        t0 = time.perf_counter()
        r = run(ctx)
        ts.append((time.perf_counter() - t0) * 1e3)
    run(prof)
    rp = run(prof)
    print("blocks %5d: %.3f ms/registration (min %.3f), %d iterations, %d passes, kernel %.1f us/pass, |t - t_gt| %.3e, T[0,3] %.9f" % (
        group, np.median(ts), min(ts), r["iterations"], r["evaluations"], rp.get("deriv_kernel_ms", 0) / max(rp["evaluations"], 1) * 1e3,
        np.linalg.norm(r["T"][:3, 3] - T_gt[:3, 3]), r["T"][0, 3]), flush=True)
    ctx.close(); prof.close()
