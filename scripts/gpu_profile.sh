#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace + stats of the default bench
# command; leaves CSV summaries under gpurun_out/<tag>/ for copying into profiles/.
#   usage: scripts/gpu_profile.sh <tag> [bench args...]
set -u
TAG=${1:-prof}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o bench -- \
    python "$ROOT/bench.py" --no-cpu-baseline "$@" > "$OUT/bench_stdout.log" 2>&1
cd "$ROOT"
find "$OUT" -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} "$OUT/kernel_stats.csv"
# keep the (large) per-dispatch trace out of the merge-back budget
find "$OUT" -name '*kernel_trace.csv' -size +20M -delete
grep '^{' "$OUT/bench_stdout.log" > "$OUT/bench_line.json" || true
head -20 "$OUT/kernel_stats.csv"
# ... and the same command with ONE registration at a time only (no in-flight phase, no host clouds, no other configs):
# the per-kernel averages of the default command's stats include launches that ran beside another context's
# (round 5: k_nn_grid 107 us "on average" against 91.8 alone)
if [ "${WM_PROFILE_SOLO:-1}" = "1" ]; then
    cd /tmp
    rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/solo" -o bench -- \
        python "$ROOT/bench.py" --no-cpu-baseline --no-other-configs --no-in-flight --no-host-clouds "$@" > "$OUT/solo_stdout.log" 2>&1
    cd "$ROOT"
    find "$OUT/solo" -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} "$OUT/kernel_stats_solo.csv"
    rm -rf "$OUT/solo"
    head -12 "$OUT/kernel_stats_solo.csv"
fi
