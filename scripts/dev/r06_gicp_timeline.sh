#!/bin/bash
# Run ON THE GPU BOX: kernel trace of the GICP config (scripts/bench_configs.py --only gicp) -> the LAST registration's
# timeline (gpurun_out/<tag>/gicp_timeline.csv: kernel, start offset, duration, gap to the previous kernel's end)
TAG=${1:-gicp_tl}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/raw" -o t -- \
    python "$ROOT/scripts/bench_configs.py" --only gicp --reps 2 > "$OUT/stdout.log" 2>&1
cd "$ROOT"
f=$(find "$OUT/raw" -name '*kernel_trace.csv' | head -1)
python3 - "$f" "$OUT/gicp_timeline.csv" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n): return re.sub(r"\(.*", "", n).replace("wm::", "")[:60]
# registrations start with k_pack (set_source); take the last complete one
starts = [i for i, r in enumerate(rows) if "k_pack" in r["Kernel_Name"] and (i == 0 or "k_pack" not in rows[i - 1]["Kernel_Name"]) ]
# every registration has two k_pack runs (source, target): group by pairs
begin = starts[-2] if len(starts) >= 2 else starts[-1]
t0 = int(rows[begin]["Start_Timestamp"]); prev = None
with open(sys.argv[2], "w") as f:
    f.write("n,kernel,start_us,dur_us,gap_us,grid,wg\n")
    for k, r in enumerate(rows[begin:]):
        st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (st - prev) / 1e3 if prev is not None else 0.0
        f.write("%d,%s,%.2f,%.2f,%.2f,%s,%s\n" % (k, short(r["Kernel_Name"]), (st - t0) / 1e3, (en - st) / 1e3, gap, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", ""))))
        prev = en
PY
rm -rf "$OUT/raw"
grep '^{' "$OUT/stdout.log" | tail -1 | cut -c1-300
