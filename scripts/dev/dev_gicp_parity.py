"""GICP: HIP path vs the CPU oracle on noisy synthetic pairs (outer / inner iteration counts, f, pose)."""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np

import __graft_entry__ as g
g.build()
from libwave_amd import capi, synth
from oracle import oracle_py as O
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
from helpers import pose_error

ctx = capi.Context(0)
for n, seed in ((30000, 21), (30000, 5), (60000, 42), (20000, 7)):
    ref, tgt, T_gt = synth.pair(n, seed=seed)
    ctx.set_source(ref)
    ctx.set_target(tgt)
    got = ctx.gicp_align()
    want = O.gicp_align(ref, tgt)
    dt, ang = pose_error(got["T"], want["T"])
    print("n=%d seed=%d: |dt| %.3e ang %.3e | outer %d/%d inner %d/%d f %.17g / %.17g n_corr %d/%d" % (
        n, seed, dt, ang, got["iterations"], want["iterations"], got["inner_total"], want["inner_total"],
        got["f"], want["f"], got["n_corr"], want["n_corr"]), flush=True)
    for k in (1, 2, 3):
        a = ctx.gicp_align(force_iterations=k)
        b = O.gicp_align(ref, tgt, force_iterations=k)
        d1, a1 = pose_error(a["T"], b["T"])
        print("   forced %d outer: |dt| %.3e ang %.3e inner %d/%d  f %.17g / %.17g" % (
            k, d1, a1, a["inner_total"], b["inner_total"], a["f"], b["f"]))
