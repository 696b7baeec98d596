#!/usr/bin/env python3
"""MultiMatcher-style throughput on ONE MI355X (run ON THE GPU BOX via gpurun): W worker threads,
one wm_ctx (own HIP stream) each, every worker registering its own pair of clouds over and over
-- the pattern of wave::MultiMatcher (multi_matcher.hpp:29-96: a queue of (ref, target) jobs and
n_threads matchers).  Reports registrations/s for small clouds, where one registration cannot
fill the GPU and concurrency is what buys throughput.  ctypes releases the GIL during the calls."""
import json
import os

# One process per GPU: keep numpy / torch CPU thread pools small.  Their default is one thread per
# logical CPU (256 here); the pools' spinning workers burn the container's CPU quota during set-up
# and the whole process is then throttled for tens of milliseconds somewhere in the timed region
# (cgroup cpu.stat: nr_throttled) -- seen as one 50-90 ms registration per run.
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "4")
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import __graft_entry__ as g
    g.build()
    from libwave_amd import capi, synth
    dev = torch.device("cuda", 0)
    for n in (10_000, 100_000):
        ref, tgt, T_gt = synth.pair(n, seed=42)
        d_ref, d_tgt = torch.from_numpy(ref).to(dev), torch.from_numpy(tgt).to(dev)
        for workers in (1, 2, 4, 8, 16):
            ctxs = [capi.Context(0) for _ in range(workers)]
            reps = 20
            errs = []

            def job(c):
                for _ in range(reps):
                    c.set_source(d_ref)
                    c.set_target(d_tgt)
                    r = c.icp_align(max_corr=3.0, max_iter=100, t_eps=1e-8, fit_eps=1e-2, carry_state=0)
                errs.append(float(np.linalg.norm(r["T"][:3, 3] - T_gt[:3, 3])) if r["T"] is not None else -1.0)

            for c in ctxs:  # warm-up (allocations, first-touch)
                c.set_source(d_ref); c.set_target(d_tgt); c.icp_align(max_corr=3.0, carry_state=0)
            torch.cuda.synchronize()
            th = [threading.Thread(target=job, args=(c,)) for c in ctxs]
            t0 = time.perf_counter()
            for t in th:
                t.start()
            for t in th:
                t.join()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(json.dumps({"points": n, "workers": workers, "registrations_per_s": workers * reps / dt,
                              "ms_per_registration_per_worker": dt / reps * 1e3,
                              "max_translation_error_m": max(errs)}))
            del ctxs


if __name__ == "__main__":
    main()
