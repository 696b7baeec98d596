#!/bin/bash
# k_gicp_cov's duration by cloud size, from a kernel trace (8 launches per size: 4 registrations x 2 clouds)
out=$PWD/gpurun_out/${1:-cov_rounds}
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out/prof -o p -- python scripts/dev/dev_cov_rounds.py > $out/run.log 2>&1
python - $out <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/prof/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_gicp_cov" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sizes = (327680, 400000, 500000, 655360, 800000, 983040)
for k, n in enumerate(sizes):
    d = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows[8 * k + 2: 8 * k + 8])
    if not d: break
    wg = (n + 255) // 256
    print("n %7d  workgroups %5d  rounds(1280) %.2f  k_gicp_cov median %.1f us  %.3f ns/point" % (n, wg, wg / 1280.0, d[len(d) // 2], d[len(d) // 2] * 1e3 / n))
PY
