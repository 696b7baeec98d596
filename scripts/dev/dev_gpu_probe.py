"""Developer probe (not a test): quick timing of the ICP path on the GPU box."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as g
g.build()
from libwave_amd import capi, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
cells = [float(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0.0]
ref, tgt, T_gt = synth.pair(n, seed=42, mode="resample")
ctx = capi.Context(0)
for h in cells:
    ctx.set_grid_cell(h)
    t0 = time.time(); ctx.set_source(ref); t1 = time.time(); ctx.set_target(tgt); t2 = time.time()
    for rep in range(3):
        r = ctx.icp_align(max_corr=3.0, force_iterations=iters, nn_method=capi.WM_NN_GRID, profile=2)
    print("  per-iter nn us:", " ".join("%.0f" % (x*1e3) for x in ctx.iteration_times()))
    r2 = ctx.icp_align(max_corr=3.0, force_iterations=iters, nn_method=capi.WM_NN_GRID, profile=0)
    print("cell=%.3f (used %.3f) levels=%d set_source %.1f ms set_target %.1f ms" % (h, r["grid_cell"], r["nn_levels"], (t1-t0)*1e3, (t2-t1)*1e3))
    print("  profile: align %.2f ms nn %.2f coarse %.2f stats %.2f solve %.2f ms; per-iter nn %.1f us; deferred %d settled %d" % (
        r["align_ms"], r["nn_ms"], r["coarse_ms"], r["stats_ms"], r["solve_ms"], r["nn_ms"]/max(r["nn_launches"],1)*1e3, r["deferred"], r.get("settled", -1)))
    print("  noprofile: align %.2f ms -> %.1f us/iter; err vs gt %s" % (r2["align_ms"], r2["align_ms"]/iters*1e3,
          np.abs(r2["T"]-T_gt).max()))
    for k in range(3):
        _, _, ms = ctx.nn_search(r2["T"], 3.0, capi.WM_NN_GRID, want=False, timed=True)
    print("  converged-pose nn kernel: %.1f us" % (ms*1e3))
    _, _, ms = ctx.nn_search(np.eye(4), 3.0, capi.WM_NN_GRID, want=False, timed=True)
    print("  identity-pose nn kernel: %.1f us" % (ms*1e3))
