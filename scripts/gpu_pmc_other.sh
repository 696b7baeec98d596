#!/bin/bash
# Run ON THE GPU BOX (via gpurun): HBM traffic counters (FETCH_SIZE / WRITE_SIZE, one pass each) of the
# GICP objective kernel and the NDT derivative kernels (BASELINE configs[2] and [3],
# scripts/bench_configs.py).  GICP's evaluations are LAUNCHED for this (WM_TUNE_GICP_SERVED=0): the
# resident evaluator is one long kernel per minimisation and reads its pairs from HBM once.
#   usage: scripts/gpu_pmc_other.sh <tag>      -> gpurun_out/<tag>/{fetch,write}_summary.csv
set -u
TAG=${1:-pmc_other}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {  # name, counters...
    local name=$1; shift
    WM_TUNE_GICP_SERVED=0 timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -o p -- \
        python "$ROOT/scripts/bench_configs.py" --reps 1 > "$OUT/$name.log" 2>&1
    find "$OUT/$name" -name '*counter_collection.csv' | head -1 | xargs -I{} cp {} "$OUT/${name}_counters.csv"
    rm -rf "$OUT/$name"
}
run fetch FETCH_SIZE
run write WRITE_SIZE
cd "$ROOT"
python3 - "$OUT" <<'PY'
import csv, sys, collections, os
out = sys.argv[1]
for name in ("fetch", "write"):
    path = os.path.join(out, name + "_counters.csv")
    if not os.path.exists(path):
        print(name, "missing"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); seen = set()
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"].split("(")[0]
        if not any(t in k for t in ("k_gicp_fdf", "k_gicp_quad", "k_ndt_derivs", "k_gicp_cov", "k_gicp_mahal")): continue
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        key = (k, row["Dispatch_Id"])
        if key not in seen:
            seen.add(key); cnt[k] += 1
    with open(os.path.join(out, name + "_summary.csv"), "w") as f:
        f.write("kernel,dispatches,counter,total,per_dispatch\n")
        for k in sorted(agg, key=lambda k: -cnt[k]):
            for c, v in agg[k].items():
                f.write("%s,%d,%s,%.6g,%.6g\n" % (k, cnt[k], c, v, v / max(cnt[k], 1)))
    print(open(os.path.join(out, name + "_summary.csv")).read()[:1500])
    os.remove(path)
PY
