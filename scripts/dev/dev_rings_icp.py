"""Developer: the headline registration (1M<->1M, 50 forced iterations) on the 64-ring lidar sampling of the scene
instead of the area-uniform one (non-uniform density)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import __graft_entry__ as g; g.build()
from libwave_amd import capi, synth
dev = torch.device("cuda", 0)
for pattern in ("uniform", "rings"):
    ref, tgt, T = synth.pair(1_000_000, seed=42, pattern=pattern)
    d_ref, d_tgt = torch.from_numpy(ref).to(dev), torch.from_numpy(tgt).to(dev)
    ctx = capi.Context(0)
    ts = []
    for rep in range(12):
        t0 = time.perf_counter()
        ctx.set_source(d_ref); ctx.set_target(d_tgt)
        r = ctx.icp_align(max_corr=3.0, force_iterations=50)
        ts.append((time.perf_counter() - t0) * 1e3)
    err = float(np.linalg.norm(r["T"][:3, 3] - T[:3, 3]))
    print("%-8s median %.3f ms (last 8: %s)  translation error %.2e m  cert launches %s" % (pattern, float(np.median(ts[4:])), [round(t, 2) for t in ts[4:]], err, r.get("cert_launches")), flush=True)
