import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import __graft_entry__ as g; g.build()
from libwave_amd import capi, synth
import torch
ctx = capi.Context(0)
pairs = []
for k in range(8):
    ref, tgt, T = synth.pair(20000, seed=100 + k)
    pairs.append((ref, tgt))
for obj in (0, 1):
    r = ctx.gicp_batch_match(pairs * 32, objective=obj)
    r = ctx.gicp_batch_match(pairs * 32, objective=obj)
    print("objective", obj, "kernel_ms", r[0]["kernel_ms"], "iters", [x["iterations"] for x in r[:8]], "evals", [x.get("evaluations") for x in r[:8]], flush=True)
