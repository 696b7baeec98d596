"""A/B of WM_TUNE_* knobs on the uniform and the 64-ring 1M pair: usage dev_rings_ab.py "K=V,..." ..."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import torch
from libwave_amd import capi, synth
for pattern in ("uniform", "rings"):
    kw = {} if pattern == "uniform" else {"pattern": "rings"}
    ref, tgt, T_gt = synth.pair(1_000_000, seed=42, **kw)
    d_ref, d_tgt = torch.from_numpy(ref).cuda(), torch.from_numpy(tgt).cuda()
    for cfg in sys.argv[1:] or ["-"]:
        keys = []
        if cfg != "-":
            for kv in cfg.split(","):
                k, v = kv.split("=")
                os.environ["WM_TUNE_" + k] = v
                keys.append("WM_TUNE_" + k)
        ctx = capi.Context(0)
        def step(profile=0):
            ctx.set_source(d_ref); ctx.set_target(d_tgt)
            return ctx.icp_align(max_corr=3.0, force_iterations=50, nn_method=capi.WM_NN_GRID, profile=profile, carry_state=0)
        for _ in range(3): step()
        ts = []
        for _ in range(10):
            t0 = time.perf_counter(); r = step(); ts.append((time.perf_counter() - t0) * 1e3)
        r1 = step(1)
        it = ctx.iteration_times() * 1e3
        print("%-8s %-28s %.3f ms (min %.3f) cert launches %d; %s" % (pattern, cfg, np.median(ts), min(ts), r["cert_launches"],
              " ".join("%d:%.0f" % (k, it[k]) for k in (0, 1, 8, 16, 26, 30, 35, 40, 49))), flush=True)
        ctx.close()
        for k in keys:
            del os.environ[k]
