#!/bin/bash
# Run ON THE GPU BOX (via gpurun): HBM traffic of the batched GICP / NDT kernels (one registration per workgroup), one
# rocprofv3 pass per counter -- FETCH_SIZE, WRITE_SIZE -- over scripts/dev/dev_gicp_batch.py / dev_ndt_batch.py with 256
# pairs of 20k points.     usage: scripts/gpu_pmc_batched.sh <tag>
set -u
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT="$ROOT/gpurun_out/${TAG}_batched"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  GICP_PAIRS=256 GICP_POINTS=20000 timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/gicp_$c" -o p -- \
      python "$ROOT/scripts/dev/dev_gicp_batch.py" > "$OUT/gicp_$c.log" 2>&1
  NDT_PAIRS=256 NDT_POINTS=20000 timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/ndt_$c" -o p -- \
      python "$ROOT/scripts/dev/dev_ndt_batch.py" > "$OUT/ndt_$c.log" 2>&1
done
python - "$OUT" <<'PY'
import csv, glob, json, os, sys
out = sys.argv[1]
res = {}
for which in ("gicp", "ndt"):
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = glob.glob(os.path.join(out, "%s_%s" % (which, c), "**", "*counter_collection.csv"), recursive=True)
        if not f:
            continue
        tot, n = 0.0, 0
        for r in csv.DictReader(open(f[0])):
            if ("k_%s_small" % which) in r["Kernel_Name"] and r["Counter_Name"] == c:
                tot += float(r["Counter_Value"])
                n += 1
        res["%s_%s" % (which, c)] = {"dispatches": n, "sum": tot, "per_dispatch": tot / max(n, 1)}
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
PY
