#!/bin/bash
# Run ON THE GPU BOX (via gpurun): calibrate rocprofv3's FETCH_SIZE on access patterns whose memory traffic is
# known (scripts/dev/fetch_calib.hip) -> gpurun_out/<tag>/fetch_calib.json.   usage: scripts/gpu_fetch_calib.sh <tag>
set -u
TAG=${1:-calib}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o /tmp/fetch_calib "$ROOT/scripts/dev/fetch_calib.hip" > "$OUT/build.log" 2>&1 || { cat "$OUT/build.log"; exit 1; }
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/prof" -o p -- /tmp/fetch_calib > "$OUT/run.log" 2>&1
f=$(find "$OUT/prof" -name '*counter_collection.csv' | head -1)
python3 - "$f" "$OUT/fetch_calib.json" <<'PY'
import csv, sys, json, collections
per = collections.defaultdict(list)
seen = {}
for row in csv.DictReader(open(sys.argv[1])):
    if row["Counter_Name"] != "FETCH_SIZE":
        continue
    key = (row["Kernel_Name"].split("(")[0], row["Dispatch_Id"])
    seen[key] = seen.get(key, 0.0) + float(row["Counter_Value"])
order = sorted(seen, key=lambda k: int(k[1]))
# launch order per repetition: stream, gather 16 B / 128-B line, gather 16 B / 64-B sector, gather 64 B / 128-B line
names = ["stream 2 GiB (float4, coalesced)", "gather 16 B per 128-B line (2^24 lines)", "gather 16 B per 64-B sector (2^25 sectors)",
         "gather 64 B per 128-B line (2^24 lines)"]
requested = [2 * 2**30, 16 * 2**24, 16 * 2**25, 64 * 2**24]
touched64 = [2 * 2**30, 64 * 2**24, 64 * 2**25, 64 * 2**24]     # bytes if memory is fetched in 64-B sectors
touched128 = [2 * 2**30, 128 * 2**24, 128 * 2**25 / 2, 128 * 2**24]  # ... in whole 128-B lines (sectors pair up)
vals = collections.defaultdict(list)
for k, key in enumerate(order):
    vals[k % 4].append(seen[key])
out = []
for k in range(4):
    v = sorted(vals[k])[len(vals[k]) // 2] * 1024.0  # FETCH_SIZE is reported in KiB
    out.append({"pattern": names[k], "bytes_requested_by_lanes": requested[k], "FETCH_SIZE_bytes": v,
                "bytes_if_64B_sectors": touched64[k], "bytes_if_128B_lines": touched128[k],
                "factor_to_64B_sector_bytes": touched64[k] / v if v else None,
                "factor_to_128B_line_bytes": touched128[k] / v if v else None})
json.dump(out, open(sys.argv[2], "w"), indent=1)
for o in out:
    print("%-46s FETCH_SIZE %8.1f MiB | x %.2f = 64-B-sector bytes | x %.2f = 128-B-line bytes" % (
        o["pattern"], o["FETCH_SIZE_bytes"] / 2**20, o["factor_to_64B_sector_bytes"] or 0, o["factor_to_128B_line_bytes"] or 0))
PY
rm -rf "$OUT/prof"
