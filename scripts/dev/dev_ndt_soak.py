"""Soak of the NDT passes' fused hand-over (tickets reset by the last workgroups, 16-byte slots): many registrations,
every one the first one's twin (same bits), on a big and on a small pair, and two contexts in flight."""
import os, sys, threading, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from libwave_amd import capi, synth

def soak(n, res, reps, pattern=None, workers=1):
    ref, tgt, _ = synth.pair(n, seed=42, pattern=pattern) if pattern else synth.pair(n, seed=42)
    d_ref, d_tgt = torch.from_numpy(ref).cuda(), torch.from_numpy(tgt).cuda()
    bad = []
    def run(w):
        c = capi.Context(0)
        first = None
        for k in range(reps):
            c.set_source(d_ref); c.set_target(d_tgt)
            r = c.ndt_align(res=res)
            if r["rc"] != 0:
                bad.append((w, k, "rc", r["rc"])); continue
            if first is None:
                first = r
            elif not (np.array_equal(first["T"], r["T"]) and first["evaluations"] == r["evaluations"] and first["score"] == r["score"]):
                bad.append((w, k, "differs"))
        c.close()
    t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(w,)) for w in range(workers)]
    [t.start() for t in th]; [t.join() for t in th]
    print("n %d res %.1f: %d x %d registrations in %.2f s, %d bad %s" % (n, res, workers, reps, time.perf_counter() - t0, len(bad), bad[:5]), flush=True)
    return len(bad)

total = 0
total += soak(2_000_000, 0.5, 150, pattern="rings")
total += soak(20000, 1.0, 3000)
total += soak(200000, 1.0, 400, workers=2)
total += soak(2_000_000, 0.5, 60, pattern="rings", workers=2)
print("SOAK", "OK" if total == 0 else "FAILED")
