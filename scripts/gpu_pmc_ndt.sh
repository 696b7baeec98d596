#!/bin/bash
# Run ON THE GPU BOX (via gpurun): SQ counters of the NDT derivative kernels (BASELINE configs[3],
# scripts/bench_configs.py --only ndt), one rocprofv3 pass per counter group.
#   usage: scripts/gpu_pmc_ndt.sh <tag>
set -u
TAG=${1:-pmc_ndt}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {  # name, counters...
    local name=$1; shift
    timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -o p -- \
        python "$ROOT/scripts/bench_configs.py" --only ndt --reps 1 > "$OUT/$name.log" 2>&1
    find "$OUT/$name" -name '*counter_collection.csv' | head -1 | xargs -I{} cp {} "$OUT/${name}_counters.csv"
    rm -rf "$OUT/$name"
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SALU
run sq2 SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run sq3 SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAVES
run f64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64
cd "$ROOT"
python3 - "$OUT" <<'PY'
import csv, sys, collections, os
out = sys.argv[1]
for name in ("sq1", "sq2", "sq3", "f64"):
    path = os.path.join(out, name + "_counters.csv")
    if not os.path.exists(path):
        print(name, "missing"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); seen = set()
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"].split("(")[0]
        if "ndt" not in k: continue
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        key = (k, row["Dispatch_Id"])
        if key not in seen:
            seen.add(key); cnt[k] += 1
    with open(os.path.join(out, name + "_summary.csv"), "w") as f:
        f.write("kernel,dispatches,counter,total,per_dispatch\n")
        for k in sorted(agg, key=lambda k: -cnt[k]):
            for c, v in agg[k].items():
                f.write("%s,%d,%s,%.6g,%.6g\n" % (k, cnt[k], c, v, v / max(cnt[k], 1)))
    print(open(os.path.join(out, name + "_summary.csv")).read()[:2500])
    os.remove(path)
PY
