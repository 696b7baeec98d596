#!/bin/bash
# Run ON THE GPU BOX (via gpurun): everything profiles/ holds for one tag, in one call --
# kernel stats of the bench command, its PMC passes, the per-iteration PMC of the search kernel,
# and the NDT derivative kernel's SQ / f64 counters.     usage: scripts/gpu_capture_all.sh <tag>
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
bash scripts/gpu_profile.sh ${TAG}_stats > /dev/null 2>&1
bash scripts/gpu_pmc.sh ${TAG}_pmc > /dev/null 2>&1
bash scripts/gpu_pmc_detail.sh ${TAG}_detail > /dev/null 2>&1
bash scripts/gpu_pmc_ndt.sh ${TAG}_ndt > /dev/null 2>&1
bash scripts/gpu_pmc_other.sh ${TAG}_other > /dev/null 2>&1
python scripts/pmc_to_json.py gpurun_out/${TAG}_pmc gpurun_out/${TAG}_pmc/pmc_latest.json gpurun_out/${TAG}_ndt gpurun_out/${TAG}_other > /dev/null
python bench.py > gpurun_out/${TAG}_bench_line_noprof.json 2> /dev/null
# the C++ wave::MultiMatcher pool: batched (deep queue) and stream-per-worker (queue of 2 x workers)
( cd libwave_amd/host
  for n in 10000 30000; do
    BENCH_QUEUE=2048 timeout 120 ./bench_multimatcher $n 6000 1 2 4
    timeout 120 ./bench_multimatcher $n 2000 4 16
  done
  timeout 120 ./bench_multimatcher 100000 600 4 16
  # the reference's default parameters (voxel filter 0.1 m + three coarser scales) and its test configuration (one scale)
  BENCH_QUEUE=512 BENCH_RES=0.1 BENCH_MULTISCALE=3 timeout 200 ./bench_multimatcher 55000 1500 1 2 4
  BENCH_RES=0.1 BENCH_MULTISCALE=3 timeout 200 ./bench_multimatcher 55000 600 4 16
  BENCH_QUEUE=512 BENCH_RES=0.1 BENCH_MULTISCALE=0 timeout 200 ./bench_multimatcher 55000 1500 1 2 ) > gpurun_out/${TAG}_multimatcher_cpp.jsonl 2> /dev/null
python scripts/dev/dev_batch_scaled.py 128 > gpurun_out/${TAG}_batch_scaled_testscan.txt 2> /dev/null
ls gpurun_out/${TAG}_*
