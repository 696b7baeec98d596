import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from libwave_amd import capi, synth
ctx = capi.Context(0)
sizes = [20000, 12000, 30000, 8000, 20001]
pairs = [synth.pair(n, seed=700 + k, mode="resample") for k, n in enumerate(sizes)]
for kw in (dict(res=2.5), dict(res=5.0, skip_line_search=1, max_iter=30), dict(res=5.0)):
    got = ctx.ndt_batch_match([(r, t) for r, t, _ in pairs], **kw)
    for (r, t, _), g in zip(pairs, got):
        ctx.set_source(r); ctx.set_target(t)
        o = ctx.ndt_align(**kw)
        print(kw, len(r), "rc %d/%d iters %d/%d passes %d/%d voxels %d/%d score %.12g/%.12g dT %.2e" % (o["rc"], g["rc"], o["iterations"], g["iterations"], o["evaluations"], g["evaluations"], o["n_voxels"], g["n_voxels"], o["score"], g["score"], np.abs(o["T"] - g["T"]).max() if o["T"] is not None and g["T"] is not None else -1))
