"""Soak of the resident GICP evaluator: many registrations in a row (and two contexts in two threads), every
evaluation must have been served or -- for the context that did not hold the device's evaluator -- launched,
results identical throughout, no registration taking unusually long (a stuck round costs >= 100 ms)."""
import os
import sys
import threading
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np

from libwave_amd import capi, synth

N = int(os.environ.get("SOAK_N", "300"))
ref, tgt, _ = synth.pair(20000, seed=3, mode="resample")


def loop(tag, n, out):
    c = capi.Context(0)
    c.set_option("gicp_served", 1)
    first = None
    worst = 0.0
    served = partial = launched = 0
    for k in range(n):
        c.set_source(ref)
        c.set_target(tgt)
        t0 = time.perf_counter()
        r = c.gicp_align()
        dt = (time.perf_counter() - t0) * 1e3
        worst = max(worst, dt)
        assert r["rc"] == 0
        if first is None:
            first = r
        assert np.array_equal(r["T"], first["T"]) and r["f"] == first["f"], (tag, k)
        if r["served_evaluations"] == r["evaluations"]:
            served += 1
        elif r["served_evaluations"] == 0:
            launched += 1
        else:
            partial += 1
    c.close()
    out[tag] = dict(worst_ms=worst, served=served, partly_served=partial, launched=launched)


out = {}
t0 = time.perf_counter()
loop("single", N, out)
print("1 thread: %.0f registrations/s" % (N / (time.perf_counter() - t0)))
for nthreads in (2, 4, 8):
    t0 = time.perf_counter()
    ts = [threading.Thread(target=loop, args=("%d threads, #%d" % (nthreads, k), N // 4, out)) for k in range(nthreads)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    print("%d threads: %.0f registrations/s" % (nthreads, nthreads * (N // 4) / (time.perf_counter() - t0)))
for k, v in out.items():
    print(k, v)
