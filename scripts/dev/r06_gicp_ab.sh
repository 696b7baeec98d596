#!/bin/bash
# Run ON THE GPU BOX: the NDT config (scripts/bench_configs.py --only gicp) under a list of environment settings, with the
# kernel stats of the model build.   usage: scripts/dev/r06_ndt_ab.sh <tag> "ENV=a" "-" ...
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
mkdir -p "$ROOT/gpurun_out/$TAG"
for setting in "$@"; do
    name=$(echo "$setting" | tr ' =' '__'); [ "$setting" = "-" ] && { setting=""; name=default; }
    cd "$ROOT"
    for rep in 1 2; do
        env $setting timeout 120 python scripts/bench_configs.py --only gicp --reps 6 2>/dev/null | grep '^{' | tail -1 | \
            python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-34s' % '$name', d['ms_per_registration'], d['outer_iterations'], d['ms_each'])"
    done
    cd /tmp && export TMPDIR=/tmp
    env $setting timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/$TAG/$name" -o t -- \
        python "$ROOT/scripts/bench_configs.py" --only gicp --reps 3 > /dev/null 2>&1
    f=$(find "$ROOT/gpurun_out/$TAG/$name" -name '*kernel_stats.csv' | head -1)
    python3 -c "import csv,sys; [print(\"    %-46s %4s calls avg %8.1f us\" % (r[\"Name\"][:46], r[\"Calls\"], float(r[\"AverageNs\"])/1e3)) for r in csv.DictReader(open(sys.argv[1])) if \"gicp_cov\" in r[\"Name\"] or \"k_nn_grid\" in r[\"Name\"]]" "$f"
    find "$ROOT/gpurun_out/$TAG/$name" -name '*kernel_trace.csv' -delete
done
