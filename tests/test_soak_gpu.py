"""Compact soaks of the two hand-over protocols that rest on gfx950's cache behaviour rather than on fences, kept in the
GPU suite ("keep the soak as a permanent GPU test"):
  * NDT's fused hand-over (wm_ndt.hip: rows written through, relaxed tickets reset by the last workgroups, sums to the
    host as 16-byte {value, pass} slots) -- every registration must be the first one's twin, bit for bit, alone and with
    two contexts in flight;
  * ICP's bins (wm_bins.hpp: integer limbs added by fire-and-forget atomics, read by the next kernel's plain loads, zeros
    put back by the solve) -- bins twice: identical bits; bins against rows of partial sums: same stop, 1e-9 m.
The long versions: scripts/dev/dev_ndt_soak.py, scripts/dev/dev_bins_soak.py."""
import threading

import numpy as np
import pytest
import torch

from libwave_amd import synth

pytestmark = pytest.mark.gpu


def _ndt_twins(wm, n, res, reps, workers, pattern=None):
    ref, tgt, _ = synth.pair(n, seed=42, pattern=pattern) if pattern else synth.pair(n, seed=42)
    d_ref, d_tgt = torch.from_numpy(ref).cuda(), torch.from_numpy(tgt).cuda()
    bad = []

    def run(w):
        c = wm.Context(0)
        first = None
        for k in range(reps):
            c.set_source(d_ref)
            c.set_target(d_tgt)
            r = c.ndt_align(res=res)
            if r["rc"] != 0:
                bad.append((w, k, "rc", r["rc"]))
            elif first is None:
                first = r
            elif not (np.array_equal(first["T"], r["T"]) and first["evaluations"] == r["evaluations"] and first["score"] == r["score"]):
                bad.append((w, k, "differs"))
        c.close()
    th = [threading.Thread(target=run, args=(w,)) for w in range(workers)]
    [t.start() for t in th]
    [t.join() for t in th]
    return bad


def test_ndt_fused_hand_over_soak(wm):
    assert _ndt_twins(wm, 20000, 1.0, 400, 1) == []
    assert _ndt_twins(wm, 200000, 1.0, 60, 2) == []
    assert _ndt_twins(wm, 1_000_000, 0.5, 12, 2, pattern="rings") == []


def test_icp_bins_soak(wm):
    rng = np.random.default_rng(99)
    cb, cr = wm.Context(0), wm.Context(0)
    cb.set_option("bins", 1)
    cr.set_option("bins", 0)
    try:
        for k in range(60):
            n = int(rng.choice([3000, 20000, 65536, 130001, 300000]))
            iters = int(rng.integers(3, 45))
            ref, tgt, _ = synth.pair(n, seed=500 + k, mode="resample")
            d_ref, d_tgt = torch.from_numpy(ref).cuda(), torch.from_numpy(tgt).cuda()
            outs = []
            for c in (cb, cb, cr):
                c.set_source(d_ref)
                c.set_target(d_tgt)
                outs.append(c.icp_align(max_corr=3.0, force_iterations=iters, nn_method=wm.WM_NN_GRID, carry_state=0))
            a, a2, b = outs
            assert a["rc"] == a2["rc"] == b["rc"] == 0, (k, n, iters)
            assert np.array_equal(a["T"], a2["T"]) and a["mse"] == a2["mse"], (k, n, iters)
            assert a["iterations"] == b["iterations"] and a["n_corr"] == b["n_corr"], (k, n, iters)
            assert np.abs(a["T"] - b["T"]).max() <= 1e-9, (k, n, iters)
    finally:
        cb.close()
        cr.close()
