"""Developer: phase cycles of the batched NDT kernel (WM_TRACE prints them per pair): mean over a batch."""
import os, sys, re, subprocess
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    from libwave_amd import capi, synth
    n, B = 20000, 64
    base = [synth.pair(n, seed=300 + k, mode="resample")[:2] for k in range(B)]
    ctx = capi.Context(0)
    ctx.ndt_batch_match(base, res=1.0)
    sys.exit(0)
env = dict(os.environ, WM_TRACE="1", OMP_NUM_THREADS="4")
out = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True).stderr
rows = re.findall(r"(\d+) iterations, (\d+) passes; kcycles: model (\d+), align (\d+) \(of it the passes (\d+)\)", out)
if not rows:
    print(out[-2000:])
else:
    import numpy as np
    a = np.array(rows, dtype=float)
    print("%d pairs; mean: %.1f iterations, %.1f passes; kcycles model %.0f, align %.0f of it passes %.0f -> control %.0f (%.1f per pass, %.1f per iteration); slowest pair %.0f" % (
        len(a), a[:, 0].mean(), a[:, 1].mean(), a[:, 2].mean(), a[:, 3].mean(), a[:, 4].mean(), (a[:, 3] - a[:, 4]).mean(),
        ((a[:, 3] - a[:, 4]) / a[:, 1]).mean(), ((a[:, 3] - a[:, 4]) / a[:, 0]).mean(), (a[:, 2] + a[:, 3]).max()))
