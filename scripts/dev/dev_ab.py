"""A/B of library tuning knobs on the bench workload (1M<->1M, 50 forced iterations).
usage: dev_ab.py "K1=V1,K2=V2" "K1=V3" ...   (each argument = one configuration of WM_TUNE_* env vars;
"-" = defaults).  Prints per configuration: median ms/registration (wall), the per-launch mean of
the search kernel (profile=1) and the per-class breakdown (profile=2)."""
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "4")
import numpy as np
import torch

from libwave_amd import capi, synth

N = int(os.environ.get("AB_POINTS", "1000000"))
ITERS = int(os.environ.get("AB_ITERS", "50"))
REPS = int(os.environ.get("AB_REPS", "12"))
ref, tgt, T_gt = synth.pair(N, seed=42)
d_ref, d_tgt = torch.from_numpy(ref).cuda(), torch.from_numpy(tgt).cuda()
ref_T = None
for cfg in sys.argv[1:] or ["-"]:
    keys = []
    if cfg != "-":
        for kv in cfg.split(","):
            k, v = kv.split("=")
            os.environ["WM_TUNE_" + k] = v
            keys.append("WM_TUNE_" + k)
    ctx = capi.Context(0)

    def step(profile):
        ctx.set_source(d_ref)
        ctx.set_target(d_tgt)
        return ctx.icp_align(max_corr=3.0, force_iterations=ITERS, nn_method=capi.WM_NN_GRID,
                             profile=profile, carry_state=0)
    for _ in range(3):
        r = step(0)
    torch.cuda.synchronize()
    ts = []
    for _ in range(REPS):
        t0 = time.perf_counter()
        r = step(0)
        ts.append((time.perf_counter() - t0) * 1e3)
    r1 = step(1)
    r2 = step(2)
    T = r["T"] if r["T"] is not None else np.zeros((4, 4))
    if ref_T is None:
        ref_T = T
    dT = float(np.abs(T - ref_T).max())
    print("%-40s ms/reg median %.3f min %.3f | align_ms %.3f | nn/launch %.1f us | p2: nn %.1f stats %.1f "
          "solve %.1f us/iter | dT vs first %.2e | err_t %.2e" % (
              cfg, float(np.median(ts)), min(ts), r["align_ms"], r1["nn_ms"] / max(r1["nn_launches"], 1) * 1e3,
              r2["nn_ms"] / ITERS * 1e3, r2["stats_ms"] / ITERS * 1e3, r2["solve_ms"] / ITERS * 1e3, dT,
              float(np.linalg.norm(T[:3, 3] - T_gt[:3, 3]))), flush=True)
    it = ctx.iteration_times() * 1e3
    step(1)
    it = ctx.iteration_times() * 1e3
    print('    late: %d iterations in %d launches, %.1f us per iteration inside' % (
        r1.get("late_iterations", 0), r1.get("late_launches", 0),
        r1.get("late_ms", 0.0) * 1e3 / max(r1.get("late_iterations", 0), 1)), flush=True)
    print('    cert launches %d; nn us by iteration:' % r.get("cert_launches", -1), ' '.join('%d:%.0f' % (k, it[k]) for k in range(len(it))), flush=True)
    c = ctx.solve_cycles()
    print('    solve kernel cycles: rows+stage %d, expand %d, pre-svd %d, svd %d, rest-of-apply %d' % (
        c[1] - c[0], c[2] - c[1], c[4] - c[2], c[5] - c[4], c[3] - c[5]), flush=True)
    ctx.close()
    for k in keys:
        del os.environ[k]
