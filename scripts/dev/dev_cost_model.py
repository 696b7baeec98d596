"""How much of the search kernel's time is lane divergence, and what would binning the queries buy?
Records the per-query work of every iteration (wm_debug_cost_log), fits
    t_iteration = a * sum_waves max_lane(trips) + b * sum_waves max_lane(chunks) + c
to the measured per-iteration kernel times, and evaluates the same model with the queries of each
2048-query chunk re-ordered by (i) their own cost (ideal), (ii) the previous iteration's cost."""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "4")
import numpy as np
import torch

from libwave_amd import capi, synth

N = int(os.environ.get("AB_POINTS", "1000000"))
ITERS = 50
ref, tgt, T_gt = synth.pair(N, seed=42)
d_ref, d_tgt = torch.from_numpy(ref).cuda(), torch.from_numpy(tgt).cuda()
ctx = capi.Context(0)


def run(profile):
    ctx.set_source(d_ref)
    ctx.set_target(d_tgt)
    return ctx.icp_align(max_corr=3.0, force_iterations=ITERS, nn_method=capi.WM_NN_GRID, profile=profile,
                         carry_state=0)


for _ in range(2):
    run(0)
r = run(1)
t_us = ctx.iteration_times() * 1e3  # production kernel, per iteration
ctx.set_source(d_ref)
ctx.set_target(d_tgt)
n = ctx.sizes()[0]
ctx.cost_log_arm(ITERS)
ctx.icp_align(max_corr=3.0, force_iterations=ITERS, nn_method=capi.WM_NN_GRID, profile=0, carry_state=0)
log = ctx.cost_log_fetch(ITERS, n)
ph = ctx.phase_log_fetch(ITERS).astype(np.float64)
print("phase cycles per wave (instrumented kernel): it  prologue passloop [walk rounds/walk walks] coop+stores tail")
for i in (0, 1, 3, 8, 14, 30, 49):
    w = max(ph[i, 7], 1)
    print("  %2d  %6.0f %7.0f [%6.0f %.2f %.2f] %6.0f %6.0f" % (i, ph[i, 3] / w, ph[i, 4] / w, ph[i, 0] / w, ph[i, 1] / max(ph[i, 2], 1), ph[i, 2] / w, ph[i, 5] / w, ph[i, 6] / w))
print("iterations logged", log.shape)
trips = (log & 0xFFFF).astype(np.int64)
chunks = ((log >> 16) & 0xFF).astype(np.int64)
passes = ((log >> 24) & 0x7F).astype(np.int64)
heavy = (log >> 31).astype(np.int64)
pad = (-n) % 2048


def waves(a):  # [n] -> [n_waves, 64], padded with zeros
    return np.pad(a, (0, pad)).reshape(-1, 64)


def model_terms(tr, ch):
    return waves(tr).max(1).sum(), waves(ch).max(1).sum()


def reorder(key):  # stable sort inside every 2048-query chunk
    k = np.pad(key, (0, pad), constant_values=0).reshape(-1, 2048)
    order = np.argsort(k, axis=1, kind="stable") + np.arange(k.shape[0])[:, None] * 2048
    return order.reshape(-1)


A = np.array([model_terms(trips[i], chunks[i]) for i in range(ITERS)], dtype=np.float64)
X = np.c_[A, np.ones(ITERS)]
coef, *_ = np.linalg.lstsq(X[1:], t_us[1:ITERS], rcond=None)  # iteration 0 (unseeded) left out of the fit
pred = X @ coef
print("fit: t_us = %.3e * sum_max_trips + %.3e * sum_max_chunks + %.2f ; residual rms %.2f us" % (
    coef[0], coef[1], coef[2], float(np.sqrt(np.mean((pred[1:] - t_us[1:ITERS]) ** 2)))))
print("it  t_us  pred | lane-eff trips chunks | passes>1  heavy | trips/q chunks/q | ideal-bin  prev-bin (pred us)")
tot = dict(cur=0.0, ideal=0.0, prev=0.0, meas=0.0)
for i in range(ITERS):
    tr, ch = trips[i], chunks[i]
    wt, wc = waves(tr), waves(ch)
    eff_t = tr.sum() / max(wt.max(1).sum() * 64, 1)
    eff_c = ch.sum() / max(wc.max(1).sum() * 64, 1)
    cost = tr + 6 * ch
    o = reorder(cost)
    trp, chp = np.pad(tr, (0, pad)), np.pad(ch, (0, pad))
    ideal = coef[0] * trp[o].reshape(-1, 64).max(1).sum() + coef[1] * chp[o].reshape(-1, 64).max(1).sum() + coef[2]
    if i > 0:
        op = reorder(trips[i - 1] + 6 * chunks[i - 1])
        prevb = coef[0] * trp[op].reshape(-1, 64).max(1).sum() + coef[1] * chp[op].reshape(-1, 64).max(1).sum() + coef[2]
    else:
        prevb = pred[i]
    tot["cur"] += pred[i]; tot["ideal"] += ideal; tot["prev"] += prevb; tot["meas"] += t_us[i]
    if i < 16 or i % 8 == 0 or i == ITERS - 1:
        print("%2d %6.1f %6.1f | %.2f %.2f | %.4f %.5f | %.2f %.2f | %6.1f %6.1f" % (
            i, t_us[i], pred[i], eff_t, eff_c, float((passes[i] > 1).mean()), float(heavy[i].mean()),
            tr.mean(), ch.mean(), ideal, prevb))
print("sum over iterations (us): measured %.0f, model %.0f, ideal binning %.0f, previous-iteration binning %.0f" % (
    tot["meas"], tot["cur"], tot["ideal"], tot["prev"]))
# distribution of the aligned state
i = ITERS - 1
print("aligned state: trips histogram", np.bincount(np.minimum(trips[i], 15))[:16].tolist())
print("aligned state: chunks histogram", np.bincount(np.minimum(chunks[i], 7))[:8].tolist())
print("aligned state: wave max trips histogram", np.bincount(np.minimum(waves(trips[i]).max(1), 20))[:21].tolist())
