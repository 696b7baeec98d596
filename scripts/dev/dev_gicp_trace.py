import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import __graft_entry__ as g
g.build()
from libwave_amd import capi, synth
from oracle import oracle_py as O
for f in ("/tmp/tg.txt", "/tmp/to.txt"):
    if os.path.exists(f): os.remove(f)
os.environ["WM_GICP_TRACE"] = "/tmp/tg.txt"
os.environ["WMO_GICP_TRACE"] = "/tmp/to.txt"
ref, tgt, T_gt = synth.pair(30000, seed=21)
ctx = capi.Context(0)
ctx.set_source(ref); ctx.set_target(tgt)
ctx.gicp_align(force_iterations=2)
O.gicp_align(ref, tgt, force_iterations=2)
a = open("/tmp/tg.txt").read().splitlines(); b = open("/tmp/to.txt").read().splitlines()
print(len(a), len(b))
for k, (x, y) in enumerate(zip(a, b)):
    if x != y:
        print("first difference at evaluation", k)
        print("gpu:", x); print("cpu:", y)
        xa, ya = x.split(), y.split()
        for i, (u, v) in enumerate(zip(xa, ya)):
            if u != v: print("  field", i, u, v)
        break
else:
    print("identical over", min(len(a), len(b)), "evaluations")
