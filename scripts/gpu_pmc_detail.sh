#!/bin/bash
# Run ON THE GPU BOX (via gpurun): per-dispatch TA / TCP / TLB counters of the correspondence
# kernel over one registration (texture-addresser and L1 pressure of the gather).
#   usage: scripts/gpu_pmc_detail.sh <tag>
set -u
TAG=${1:-pmcd}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {
    local name=$1; shift
    timeout 150 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -o p -- \
        python "$ROOT/bench.py" --no-cpu-baseline --no-other-configs --steps 1 --warmup 1 > "$OUT/$name.log" 2>&1
    find "$OUT/$name" -name '*counter_collection.csv' | head -1 | xargs -I{} cp {} "$OUT/${name}_counters.csv"
    rm -rf "$OUT/$name"
}
run sq1 SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES
run sq2 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM
run sq3 SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_RD SQ_BUSY_CYCLES SQ_INSTS_BRANCH
run ta1 TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE
run ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
run tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
run tcp2 TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
run tlb TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum
cd "$ROOT"
python3 - "$OUT" <<'PY'
import csv, sys, collections, os
out = sys.argv[1]
for name in ("sq1", "sq2", "sq3", "ta1", "ta2", "tcp1", "tcp2", "tlb"):
    path = os.path.join(out, name + "_counters.csv")
    if not os.path.exists(path):
        print(name, "missing"); continue
    per = collections.OrderedDict()
    for row in csv.DictReader(open(path)):
        kn = row["Kernel_Name"]
        if "k_nn_grid" not in kn and "k_nn_cert" not in kn: continue   # the two kernels of the correspondence step
        d = per.setdefault(int(row["Dispatch_Id"]), {})
        d[row["Counter_Name"]] = float(row["Counter_Value"])
        d["is_cert"] = 1.0 if "k_nn_cert" in kn else 0.0
    ids = sorted(per)[-50:]  # the last registration
    names = sorted(per[ids[0]]) if ids else []
    with open(os.path.join(out, name + "_per_dispatch.csv"), "w") as f:
        f.write("iteration," + ",".join(names) + "\n")
        for k, i in enumerate(ids):
            f.write("%d," % k + ",".join("%.6g" % per[i].get(n, float("nan")) for n in names) + "\n")
    print(open(os.path.join(out, name + "_per_dispatch.csv")).read()[:6000])
    os.remove(path)
PY
