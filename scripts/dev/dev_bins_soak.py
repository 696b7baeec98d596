"""Soak of the bins pipeline (libwave_amd/csrc/wm_bins.hpp): hundreds of registrations of clouds of many sizes, every one
run twice with the sums in bins (must be bit-identical: integer sums commute) and once with rows of partial sums (must
agree to 1e-9 m, stop at the same iteration).  What it would catch: a lost or doubled atomic, a stale L2 line under the
solve's plain loads, zeros not put back.      usage: python scripts/dev/dev_bins_soak.py [registrations]"""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import torch

from libwave_amd import capi as wm, synth

N_REG = int(sys.argv[1]) if len(sys.argv) > 1 else 300
THREADS = int(sys.argv[2]) if len(sys.argv) > 2 else 1   # > 1: that many soaks side by side (contexts in flight on one GPU)
import threading
results = {}


def soak(tid):
  rng = np.random.default_rng(1234 + tid)
  ctx_b = wm.Context(0)
  ctx_r = wm.Context(0)
  ctx_b.set_option("bins", 1)
  ctx_r.set_option("bins", 0)
  bad = 0
  worst = 0.0
  for k in range(N_REG):
      n = int(rng.choice([3000, 20000, 64 * 1024, 130001, 200000, 500000, 1000000], p=[.2, .2, .15, .15, .15, .1, .05]))
      iters = int(rng.integers(3, 60))
      ref, tgt, _ = synth.pair(n, seed=1000 + k + 100000 * tid, mode="resample")
      d_ref, d_tgt = torch.from_numpy(ref).cuda(), torch.from_numpy(tgt).cuda()
      outs = []
      for c in (ctx_b, ctx_b, ctx_r):
          c.set_source(d_ref)
          c.set_target(d_tgt)
          outs.append(c.icp_align(max_corr=3.0, force_iterations=iters, nn_method=wm.WM_NN_GRID, carry_state=0))
      a, a2, b = outs
      ok = a["rc"] == 0 and a2["rc"] == 0 and b["rc"] == 0 and np.array_equal(a["T"], a2["T"]) and a["mse"] == a2["mse"]
      dt = float(np.abs(a["T"] - b["T"]).max()) if ok else float("nan")
      ok = ok and dt <= 1e-9 and a["iterations"] == b["iterations"] and a["n_corr"] == b["n_corr"]
      worst = max(worst, dt if dt == dt else 0.0)
      if not ok:
          bad += 1
          print("[%d] MISMATCH at registration %d: n=%d iterations=%d rc=%s/%s/%s max|dT|=%g" % (tid, k, n, iters, a["rc"], a2["rc"], b["rc"], dt))
      if k % 50 == 49:
          print("[%d] %d registrations, %d mismatches, worst bins-vs-rows |dT| %.2e" % (tid, k + 1, bad, worst), flush=True)
  results[tid] = (bad, worst)


th = [threading.Thread(target=soak, args=(t,)) for t in range(THREADS)]
[t.start() for t in th]
[t.join() for t in th]
bad = sum(v[0] for v in results.values()) + (THREADS - len(results))
worst = max([v[1] for v in results.values()] + [0.0])
print("SOAK %s: %d x %d registrations, %d mismatches, worst bins-vs-rows |dT| %.2e" % ("OK" if bad == 0 else "FAILED", THREADS, N_REG, bad, worst))
sys.exit(0 if bad == 0 else 1)
