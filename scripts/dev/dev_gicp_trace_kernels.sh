#!/bin/bash
# Run ON THE GPU BOX: kernel trace of GICP 500k registrations (statistics objective), per-kernel totals per registration.
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/gq
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gq -o p -- python /root/repo/scripts/dev/dev_gicp_quad_time.py 500000 > /tmp/gq.log 2>&1
f=$(find /tmp/gq -name "*kernel_stats.csv" | head -1)
tail -3 /tmp/gq.log
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
# 12 registrations in the script (6 per objective): report per-kernel total / 12 is meaningless across objectives; print raw
for r in rows[:28]:
    print("%-58s calls %5s total %9.1f us avg %8.1f us %5s %%" % (r["Name"][:58], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
