"""NDT, BASELINE configs[3] (2M ring scan, res 0.5): voxels within `res` per source point, and what a wave of 64
consecutive (Morton-ordered) points pays for them -- the max over its lanes -- against the mean.  numpy / cKDTree."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from scipy.spatial import cKDTree
from libwave_amd import synth

n = int(os.environ.get("N", 2_000_000))
res = 0.5
ref, tgt, T = synth.pair(n, seed=42, pattern="rings")
tgt = tgt[:, :3].astype(np.float64); ref = ref[:, :3].astype(np.float64)
cell = np.floor(tgt / res).astype(np.int64)
key = (cell[:, 2] << 42) + ((cell[:, 1] + (1 << 20)) << 21) + (cell[:, 0] + (1 << 20))
o = np.argsort(key, kind="stable")
ks, first, cnt = np.unique(key[o], return_index=True, return_counts=True)
sums = np.add.reduceat(tgt[o], first, axis=0)
means = (sums / cnt[:, None])[cnt >= 6]
print("voxels with >= 6 points:", len(means), "of", len(ks))
tree = cKDTree(means)
def morton(c):
    c = c - c.min(0)
    def part(v):
        v = v.astype(np.uint64) & 0x1FFFFF
        v = (v | (v << 32)) & 0x1F00000000FFFF
        v = (v | (v << 16)) & 0x1F0000FF0000FF
        v = (v | (v << 8)) & 0x100F00F00F00F00F
        v = (v | (v << 4)) & 0x10C30C30C30C30C3
        v = (v | (v << 2)) & 0x1249249249249249
        return v
    return part(c[:, 0]) | (part(c[:, 1]) << 1) | (part(c[:, 2]) << 2)
m = morton(np.floor(ref / 0.18).astype(np.int64))
order = np.argsort(m, kind="stable")
pts = ref[order]
sel = np.arange(0, len(pts) - 64 * 4000, max(1, (len(pts) // (64 * 4000))) * 64)[:4000]
idx = (sel[:, None] + np.arange(64)[None, :]).ravel()
counts = tree.query_ball_point(pts[idx], res, return_length=True).reshape(-1, 64)
print("per point: mean %.2f  p50 %d  p90 %d  max %d   zero: %.1f %%" % (counts.mean(), np.median(counts), np.percentile(counts, 90), counts.max(), 100 * (counts == 0).mean()))
wmax = counts.max(1)
print("per wave of 64 Morton-consecutive points: mean of max %.2f, mean of mean %.2f  -> lanes busy in the pair loop %.1f %%" % (wmax.mean(), counts.mean(1).mean(), 100 * counts.sum() / (wmax.sum() * 64)))
# sorted by count (stable within class): waves homogeneous
flat = counts.ravel()
srt = np.sort(flat)[: len(flat) // 64 * 64].reshape(-1, 64)
print("the same points grouped by count: lanes busy %.1f %%, trips %.0f vs %.0f" % (100 * srt.sum() / (srt.max(1).sum() * 64), srt.max(1).sum(), wmax.sum()))
hist = np.bincount(flat, minlength=12)
print("histogram of counts:", hist[:14].tolist())
