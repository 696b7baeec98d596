#!/bin/bash
# Run ON THE GPU BOX (via gpurun): the GPU suite, the bench line (with the registrations-in-flight block), and the C++
# wave::MultiMatcher pool at BASELINE configs[1]'s size (1M<->1M, 50 forced iterations) with 1 / 2 / 4 workers.
#   usage: scripts/gpu_r05_first.sh <tag>
set -u
TAG=${1:-r05a}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/${TAG}_build.log 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/${TAG}_pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench.err
( cd libwave_amd/host
  # configs[1] through the C++ pool: full resolution, 50 forced iterations, estimateInfo = LUMold only / the default (LUM + Censi + LUMold)
  WAVE_ICP_BENCH_FORCE_ITERATIONS=50 BENCH_ESTIMATOR=LUM_OLD timeout 300 ./bench_multimatcher 1000000 48 1 2 4
  WAVE_ICP_BENCH_FORCE_ITERATIONS=50 timeout 300 ./bench_multimatcher 1000000 48 1 2 4
  # one protocol for the small-pair pools: >= 20 000 pairs per row, the reference's default queue of 10
  BENCH_QUEUE=10 BENCH_PAIR=copy timeout 200 ./bench_multimatcher 10000 40000 1 4 16
  BENCH_QUEUE=10 timeout 200 ./bench_multimatcher 10000 40000 1 4 16
  BENCH_MATCHER=gicp BENCH_QUEUE=10 timeout 300 ./bench_multimatcher 20000 20000 4 16
  BENCH_MATCHER=ndt BENCH_QUEUE=10 timeout 300 ./bench_multimatcher 20000 40000 4 16 ) > gpurun_out/${TAG}_pool.jsonl 2> gpurun_out/${TAG}_pool.err
tail -3 gpurun_out/${TAG}_pytest.log
python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/${TAG}_bench_line.json").read().strip().splitlines()[-1])
    print("value", d["value"], "ms", d["ms_per_step"], {k: v for k, v in d["config"].items() if "in_flight" in k or "h2d" in k})
except Exception as e:
    print("bench line unreadable:", e)
PY
cat gpurun_out/${TAG}_pool.jsonl | cut -c1-330
