#!/bin/bash
# Run ON THE GPU BOX: kernel trace of the bench command (one registration at a time) -> the timeline of the LAST timed
# registration: per dispatch name, start offset, duration, gap to the previous kernel's end (gpurun_out/<tag>/timeline.csv)
TAG=${1:-timeline}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
env "$@" timeout 200 rocprofv3 --kernel-trace --output-format csv -d "$OUT/raw" -o t -- \
    python "$ROOT/bench.py" --no-cpu-baseline --no-other-configs --no-in-flight ${TL_BENCH_FLAGS---no-host-clouds} --steps 4 --warmup 2 > "$OUT/stdout.log" 2>&1
cd "$ROOT"
f=$(find "$OUT/raw" -name '*kernel_trace.csv' | head -1)
python3 - "$f" "$OUT/timeline.csv" ${TL_WHICH:--2} <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = re.sub(r"\(.*", "", n)
    n = n.replace("wm::", "")
    return n[:60]
# registrations end with k_fix_keys + k_fetch_signal<64>; the LAST one is event-bracketed by bench.py (a barrier packet
# around every search launch): take the one before it -- it runs exactly as a caller's registration would
ends = [i for i, r in enumerate(rows) if "k_fix_keys" in r["Kernel_Name"]]
which = int(sys.argv[3]) if len(sys.argv) > 3 else -2
e = ends[which] + 1          # (+ the state's fetch)
begin = ends[which - 1] + 2
t0 = int(rows[begin]["Start_Timestamp"])
prev_end = None
with open(sys.argv[2], "w") as f:
    f.write("n,kernel,start_us,dur_us,gap_us,grid,wg\n")
    for k, r in enumerate(rows[begin:e + 1]):
        st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (st - prev_end) / 1e3 if prev_end is not None else 0.0
        f.write("%d,%s,%.2f,%.2f,%.2f,%s,%s\n" % (k, short(r["Kernel_Name"]), (st - t0) / 1e3, (en - st) / 1e3, gap, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", ""))))
        prev_end = en
print(open(sys.argv[2]).read()[:200])
PY
rm -rf "$OUT/raw"
grep '^{' "$OUT/stdout.log" | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
