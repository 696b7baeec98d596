"""First evaluation at which the batched small GICP (pair 0) and the one-pair path part ways."""
import os
import subprocess
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np

from libwave_amd import capi, synth

n = int(os.environ.get("GICP_POINTS", "5000"))
ref, tgt, T_gt = synth.pair(n, seed=int(os.environ.get("GICP_SEED", "100")), mode=os.environ.get("GICP_MODE", "copy"))
ctx = capi.Context(0)
path = "/tmp/gicp_one.trace"
if os.path.exists(path):
    os.remove(path)
os.environ["WM_GICP_TRACE"] = path
ctx.gicp_match(ref, tgt)
del os.environ["WM_GICP_TRACE"]
os.environ["WM_GICP_SMALL_TRACE"] = "1"
sys.stdout.flush()
got = ctx.gicp_batch_match([(ref, tgt)])
