"""One-pair NDT (wm_ndt_align) on SMALL clouds: the per-pass overheads (launch, sums' hand-over) are all there is."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from libwave_amd import capi, synth
for n, res in ((20000, 1.0), (55000, 5.0), (200000, 1.0)):
    ref, tgt, T_gt = synth.pair(n, seed=7)
    d_ref, d_tgt = torch.from_numpy(ref).cuda(), torch.from_numpy(tgt).cuda()
    for fused in ("1", "0"):
        os.environ["WM_TUNE_NDT_FUSED_FETCH"] = fused
        c = capi.Context(0)
        def run():
            c.set_source(d_ref); c.set_target(d_tgt)
            return c.ndt_align(res=res)
        run(); run()
        ts = []
        for _ in range(15):
            t0 = time.perf_counter(); r = run(); ts.append((time.perf_counter() - t0) * 1e3)
        print("n %6d res %.1f fused %s: %.3f ms/registration, %d passes -> %.1f us per pass all in" % (n, res, fused, np.median(ts), r["evaluations"], np.median(ts) * 1e3 / max(r["evaluations"], 1)), flush=True)
        c.close()
