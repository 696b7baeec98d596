"""What the host steers the search-kernel choice by (step size, changed matches, searched share), per
iteration, on the uniform and the 64-ring sampling of the bench scene -- and what each kernel costs there."""
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import torch

from libwave_amd import capi, synth

for pattern in ("uniform", "rings"):
    kw = {} if pattern == "uniform" else {"pattern": "rings"}
    ref, tgt, T_gt = synth.pair(1_000_000, seed=42, **kw)
    d_ref, d_tgt = torch.from_numpy(ref).cuda(), torch.from_numpy(tgt).cuda()
    logs = {}
    for name, cert_from in (("full", -2), ("cert0", 0), ("auto", -1)):
        ctx = capi.Context(0)
        ctx.set_option("cert_from", cert_from)

        def step(profile=0):
            ctx.set_source(d_ref)
            ctx.set_target(d_tgt)
            return ctx.icp_align(max_corr=3.0, force_iterations=50, nn_method=capi.WM_NN_GRID, profile=profile, carry_state=0)
        for _ in range(2):
            step()
        ts = []
        for _ in range(6):
            t0 = time.perf_counter()
            r = step()
            ts.append((time.perf_counter() - t0) * 1e3)
        r1 = step(1)
        logs[name] = (np.median(ts), r1["cert_launches"], ctx.iteration_times() * 1e3, ctx.pub_log())
        ctx.close()
    print("== %s: full %.3f ms | cert from 0 %.3f ms | auto %.3f ms (%d certificate launches)" % (
        pattern, logs["full"][0], logs["cert0"][0], logs["auto"][0], logs["auto"][1]))
    print(" it | step mm | changed %% (full) | searched %% (cert0) | us: full cert0 auto")
    for k in range(50):
        pf, pc = logs["full"][3], logs["cert0"][3]
        print(" %2d | %7.2f | %6.2f | %6.2f | %5.0f %5.0f %5.0f" % (
            k, pf[k][1] * 1e3 if k < len(pf) else -1, pf[k][2] * 100 if k < len(pf) else -1,
            pc[k][3] * 100 if k < len(pc) else -1, logs["full"][2][k], logs["cert0"][2][k], logs["auto"][2][k]))
