#!/usr/bin/env python3
"""BASELINE.json configs[2] and configs[3] on one MI355X (run ON THE GPU BOX via gpurun):
  configs[2]  GICPMatcher 500k<->500k clouds with per-point covariances
  configs[3]  NDTMatcher 2M-point scan, 0.5 m voxel grid
Each is one full C-ABI registration with both clouds already resident in HBM
(wm_set_source + wm_set_target + wm_gicp_align / wm_ndt_align), timed after warm-up.
Prints one JSON line per config; `python bench.py` stays the headline (configs[1])."""
import json
import os
import sys
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
    ctx = capi.Context(0)
    reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 5
    only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else None  # gicp | ndt

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            r = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3, r

    if only in (None, "gicp"):
        run_gicp(ctx, dev, timed, synth, torch)
    if only in (None, "ndt"):
        run_ndt(ctx, dev, timed, synth, torch)


def run_gicp(ctx, dev, timed, synth, torch):
    ref, tgt, T_gt = synth.pair(500_000, seed=42)
    d_ref, d_tgt = torch.from_numpy(ref).to(dev), torch.from_numpy(tgt).to(dev)

    def gicp():
        ctx.set_source(d_ref)
        ctx.set_target(d_tgt)
        return ctx.gicp_align()
    ms, r = timed(gicp)
    err = float(np.linalg.norm(r["T"][:3, 3] - T_gt[:3, 3])) if r["T"] is not None else None
    print(json.dumps({"config": "GICPMatcher 500k<->500k (BASELINE configs[2])", "ms_per_registration": ms,
                      "registrations_per_s": 1e3 / ms, "rc": r["rc"],
                      "outer_iterations": r.get("iterations"), "translation_error_m": err,
                      "detail": {k: v for k, v in r.items() if k not in ("T",) and np.isscalar(v)}}))



def run_ndt(ctx, dev, timed, synth, torch):
    ref, tgt, T_gt = synth.pair(2_000_000, seed=42)
    d_ref, d_tgt = torch.from_numpy(ref).to(dev), torch.from_numpy(tgt).to(dev)

    def ndt():
        ctx.set_source(d_ref)
        ctx.set_target(d_tgt)
        return ctx.ndt_align(res=0.5)
    ms, r = timed(ndt)
    err = float(np.linalg.norm(r["T"][:3, 3] - T_gt[:3, 3])) if r["T"] is not None else None
    print(json.dumps({"config": "NDTMatcher 2M<->2M, 0.5 m voxels (BASELINE configs[3])",
                      "ms_per_registration": ms, "registrations_per_s": 1e3 / ms, "rc": r["rc"],
                      "iterations": r["iterations"], "n_voxels": r["n_voxels"],
                      "derivative_passes": r["evaluations"], "translation_error_m": err}))


if __name__ == "__main__":
    main()
