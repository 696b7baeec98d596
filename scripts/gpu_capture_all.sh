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
# the C++ wave::MultiMatcher pool: at the REFERENCE'S DEFAULT queue of 10 (multi_matcher.hpp:32-34) with 1-16 workers
# (steady state: 24 000 pairs, so that every worker's staging buffers have grown before most of the run), with a
# deep queue, and the voxel-filtered configurations
( cd libwave_amd/host
  BENCH_QUEUE=10 timeout 200 ./bench_multimatcher 10000 24000 1 2 4 8 16
  BENCH_QUEUE=2048 timeout 200 ./bench_multimatcher 10000 24000 1 2 4 16
  BENCH_QUEUE=10 timeout 200 ./bench_multimatcher 30000 9000 1 4 16
  BENCH_QUEUE=2048 timeout 200 ./bench_multimatcher 30000 9000 1 4 16
  BENCH_QUEUE=10 timeout 120 ./bench_multimatcher 100000 900 4 16
  # the reference's default parameters (voxel filter 0.1 m + three coarser scales) and its test configuration (one scale)
  BENCH_QUEUE=10 BENCH_RES=0.1 BENCH_MULTISCALE=3 timeout 200 ./bench_multimatcher 55000 3000 1 4 16
  BENCH_QUEUE=512 BENCH_RES=0.1 BENCH_MULTISCALE=3 timeout 200 ./bench_multimatcher 55000 3000 1 2 4
  BENCH_QUEUE=10 BENCH_RES=0.1 BENCH_MULTISCALE=0 timeout 200 ./bench_multimatcher 55000 3000 1 4
  # (steady state: 16 000 / 40 000 pairs -- a launch of 256 lasts 25 / 12 ms, a run of 4 000 pairs is warm-up and tail)
  # MultiMatcher<GICPMatcher>: the batched small GICP (one registration per compute unit), full resolution and the
  # matcher's default voxel filter; MultiMatcher<NDTMatcher>: the batched small NDT, 1 m voxels and the matcher's default 5 m
  BENCH_MATCHER=gicp BENCH_QUEUE=10 timeout 200 ./bench_multimatcher 20000 20000 1 2 4 8 16
  BENCH_MATCHER=gicp BENCH_QUEUE=2048 timeout 200 ./bench_multimatcher 20000 20000 1 2 4 16
  BENCH_MATCHER=gicp BENCH_QUEUE=10 timeout 200 ./bench_multimatcher 10000 6000 4 16
  BENCH_MATCHER=gicp BENCH_QUEUE=10 BENCH_RES=0.1 timeout 200 ./bench_multimatcher 55000 1500 4 16
  BENCH_MATCHER=ndt BENCH_QUEUE=10 timeout 200 ./bench_multimatcher 20000 40000 1 2 4 8 16
  BENCH_MATCHER=ndt BENCH_QUEUE=2048 timeout 200 ./bench_multimatcher 20000 40000 1 2 4 16
  BENCH_MATCHER=ndt BENCH_QUEUE=10 BENCH_RES=5 timeout 200 ./bench_multimatcher 55000 2000 4 16 ) > gpurun_out/${TAG}_multimatcher_cpp.jsonl 2> /dev/null
python scripts/dev/dev_batch_scaled.py 128 > gpurun_out/${TAG}_batch_scaled_testscan.txt 2> /dev/null
ls gpurun_out/${TAG}_*
