#!/bin/bash
# Run ON THE GPU BOX (via gpurun): per-dispatch TA / TCP / TLB counters of the correspondence
# kernel over one registration (texture-addresser and L1 pressure of the gather).
# The bench command is run WITHOUT its in-flight / host-cloud / other-config phases, so that the process's last 50
# dispatches of the two search kernels ARE the one timed registration (round 5's captures took the last 50 of a run
# that ended with interleaved contexts: 49 certificate launches of several registrations).  The rows are checked
# against what a registration looks like -- full searches first (15 625 waves at 1M points, the first one the most
# expensive), then certificate launches (3 9xx waves), as many of each as the bench line says -- and the script FAILS
# (exit 3, no csv) when they do not.
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
        python "$ROOT/bench.py" --no-cpu-baseline --no-other-configs --no-in-flight --no-host-clouds --steps 1 --warmup 1 > "$OUT/$name.log" 2>&1
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
    line = {}
    try:
        import json
        for l in open(os.path.join(out, name + ".log")):
            if l.startswith("{"):
                line = json.loads(l)
    except Exception:
        pass
    iters = int(line.get("config", {}).get("iterations", 50))
    want_cert = line.get("config", {}).get("cert_launches_per_registration")
    ids = sorted(per)[-iters:]  # the last registration = the timed step (nothing runs behind it: see the flags above)
    names = sorted(per[ids[0]]) if ids else []
    # ---- is this one registration?
    seq = [int(per[i]["is_cert"]) for i in ids]
    problems = []
    if len(ids) != iters:
        problems.append("%d dispatches of the search kernels, wanted %d" % (len(ids), iters))
    if seq and seq[0] != 0:
        problems.append("iteration 0 is not a full search")
    if want_cert is not None and sum(seq) != int(want_cert):
        problems.append("%d certificate launches, the bench line says %s" % (sum(seq), want_cert))
    if "SQ_WAVES" in names:
        for k, i in enumerate(ids):
            w = per[i]["SQ_WAVES"]
            ok = (3000 <= w <= 4200) if seq[k] else (15000 <= w <= 16500)
            if not ok and int(line.get("config", {}).get("points_per_gpu", 1000000)) == 1000000:
                problems.append("iteration %d: %s with %d waves" % (k, "k_nn_cert" if seq[k] else "k_nn_grid", int(w)))
                break
    if "SQ_INSTS_VALU" in names and ids:
        v = [per[i]["SQ_INSTS_VALU"] for i in ids]
        if v[0] < max(v) * 0.999:
            problems.append("iteration 0 is not the most expensive launch (%.3g vs max %.3g VALU): not an unseeded search" % (v[0], max(v)))
    if problems:
        print("NOT ONE REGISTRATION (%s): %s" % (name, "; ".join(problems)))
        sys.exit(3)
    with open(os.path.join(out, name + "_per_dispatch.csv"), "w") as f:
        f.write("iteration," + ",".join(names) + "\n")
        for k, i in enumerate(ids):
            f.write("%d," % k + ",".join("%.6g" % per[i].get(n, float("nan")) for n in names) + "\n")
    print(open(os.path.join(out, name + "_per_dispatch.csv")).read()[:6000])
    os.remove(path)
PY
