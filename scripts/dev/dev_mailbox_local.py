"""Two emulated ranks on one GPU exchanging through mailboxes (WM_COMM_P2P_LOCAL=1): what stalls?
(1) the exchange kernel alone, both ranks at once; (2) a sharded registration on pre-warmed contexts."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["WM_COMM_P2P_LOCAL"] = "1"
os.environ["WM_COMM_P2P_TIMEOUT_MS"] = os.environ.get("WM_COMM_P2P_TIMEOUT_MS", "2000")
import numpy as np
from libwave_amd import capi as wm, synth

world = int(os.environ.get("WORLD", "2"))
ref, tgt, _ = synth.pair(40000, seed=17)
ctxs = [wm.Context(0) for _ in range(world)]
for c in ctxs:  # warm: every buffer of a registration exists
    c.set_source(ref); c.set_target(tgt)
    c.icp_align(max_corr=3.0, force_iterations=3, nn_method=wm.WM_NN_GRID)
comms = wm.Comm.init_local(world, 0)

def both(fn):
    outs = [None] * world
    def run(r):
        try:
            outs[r] = fn(r)
        except Exception as e:
            outs[r] = "ERR " + str(e)[:200]
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    t0 = time.perf_counter()
    [t.start() for t in th]; [t.join() for t in th]
    return outs, time.perf_counter() - t0

o, dt = both(lambda r: ctxs[r].allreduce_probe(comms[r], reps=20))
print("probe alone:", o, "%.2f s" % dt, flush=True)
for k in range(3):
    o, dt = both(lambda r: ctxs[r].icp_align_sharded(comms[r], ref, tgt, max_corr=3.0, force_iterations=15, nn_method=wm.WM_NN_GRID))
    print("registration %d: %.2f s" % (k, dt), [x if isinstance(x, str) else (x["rc"], x["exchange_in_kernel"], x["iterations"]) for x in o], flush=True)
    if all(not isinstance(x, str) for x in o):
        print("  same bits on both ranks:", all(np.array_equal(o[0]["T"], x["T"]) for x in o))
