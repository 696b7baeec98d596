#!/bin/bash
# Run ON THE GPU BOX: per-kernel time of one developer script (kernel-trace + stats), top lines only.
#   bash scripts/dev/prof_stats.sh scripts/dev/dev_cov_time.py [lines]
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_stats
WM_TUNE_GICP_SERVED=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o p -- python "/root/repo/$1" > /tmp/prof_stats.log 2>&1
f=$(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1)
if [ -z "$f" ]; then echo "no stats file"; tail -5 /tmp/prof_stats.log; exit 1; fi
python - "$f" "${2:-25}" <<'PY'
import csv, sys
for i, r in enumerate(csv.DictReader(open(sys.argv[1]))):
    if i >= int(sys.argv[2]): break
    print("%-60s calls %5s avg %10.1f us  %5s %%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
