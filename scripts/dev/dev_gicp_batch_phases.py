"""Developer: phase cycles of the batched GICP kernel (WM_TRACE prints them per pair): mean over a batch."""
import os, sys, re, subprocess
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import numpy as np
    from libwave_amd import capi, synth
    obj = int(sys.argv[2])
    n, B = 20000, 64
    base = [synth.pair(n, seed=300 + k, mode="resample")[:2] for k in range(B)]
    ctx = capi.Context(0)
    ctx.gicp_batch_match(base, objective=obj)
    sys.exit(0)
for obj in (0, 1):
    env = dict(os.environ, WM_TRACE="1", OMP_NUM_THREADS="4")
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(obj)], env=env, capture_output=True, text=True).stderr
    rows = re.findall(r"kcycles: grids (\d+), covariances (\d+), searches (\d+), minimisations (\d+)", out)
    if not rows:
        print(out[-2000:])
        continue
    import numpy as np
    a = np.array(rows, dtype=float)
    print("objective %d: %d pairs; mean kcycles grids %.0f covariances %.0f searches(+statistics) %.0f minimisations %.0f | max total %.0f" % (
        obj, len(a), *a.mean(0), a.sum(1).max()))
