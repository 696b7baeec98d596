#!/bin/bash
# Run ON THE GPU BOX (via gpurun): PMC counters for the bench command, one rocprofv3 pass
# per counter group (TCC FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950).
#   usage: scripts/gpu_pmc.sh <tag> [bench args...]
set -u
TAG=${1:-pmc}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {  # name, counters...
    local name=$1; shift
    rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -o p -- \
        python "$ROOT/bench.py" --no-cpu-baseline --no-other-configs --no-in-flight --steps 2 --warmup 1 > "$OUT/$name.log" 2>&1
    find "$OUT/$name" -name '*counter_collection.csv' | head -1 | xargs -I{} cp {} "$OUT/${name}_counters.csv"
    rm -rf "$OUT/$name"
}
run sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_SALU SQ_BUSY_CYCLES
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum
cd "$ROOT"
python3 - "$OUT" <<'PY'
import csv, sys, collections, os
out = sys.argv[1]
for name in ("sq", "fetch", "write", "tcc"):
    path = os.path.join(out, name + "_counters.csv")
    if not os.path.exists(path):
        print(name, "missing"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    seen = set()
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"].split("(")[0]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        key = (k, row["Dispatch_Id"])
        if key not in seen:
            seen.add(key); cnt[k] += 1
    with open(os.path.join(out, name + "_summary.csv"), "w") as f:
        f.write("kernel,dispatches,counter,total,per_dispatch\n")
        for k in sorted(agg, key=lambda k: -cnt[k]):
            for c, v in agg[k].items():
                f.write("%s,%d,%s,%.6g,%.6g\n" % (k, cnt[k], c, v, v / max(cnt[k], 1)))
    print(open(os.path.join(out, name + "_summary.csv")).read()[:1800])
    if os.path.getsize(path) > 8 << 20:
        os.remove(path)
PY
