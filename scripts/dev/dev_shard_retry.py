import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
from libwave_amd import capi as wm, synth
ref, tgt, T = synth.pair(60000, seed=21)
for shift in (2.0, 4.0, 6.0):
    r2 = ref.copy(); r2[:, 0] -= shift
    c = wm.Context(0); c.set_source(r2); c.set_target(tgt)
    want = c.icp_align(max_corr=3.0, force_iterations=30, nn_method=wm.WM_NN_GRID); c.close()
    for world in (2, 4):
        m = wm.Multi([0] * world, emulate=True)
        got = m.icp_align(r2, tgt, max_corr=3.0, force_iterations=30, nn_method=wm.WM_NN_GRID)
        m.close()
        print(shift, world, "attempts", got["shard_attempts"], "violations", got["owned_violations"], "rc", got["rc"], want["rc"],
              "dT", None if got["T"] is None or want["T"] is None else float(np.abs(got["T"] - want["T"]).max()), "tx", None if want["T"] is None else want["T"][0, 3])
