#!/bin/bash
# Run ON THE GPU BOX (via gpurun): the counter and kernel-stats captures of scripts/gpu_capture_all.sh WITHOUT its
# MultiMatcher pool sweeps (those take ten minutes): what bench.py's roofline block reads (profiles/pmc_latest.json)
# and the per-iteration counters of one registration.      usage: scripts/gpu_capture_core.sh <tag>
set -u
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
bash scripts/gpu_profile.sh ${TAG}_stats > /dev/null 2>&1
bash scripts/gpu_pmc.sh ${TAG}_pmc > /dev/null 2>&1
bash scripts/gpu_pmc_detail.sh ${TAG}_detail > gpurun_out/${TAG}_detail.log 2>&1; echo "pmc_detail rc=$?"
bash scripts/gpu_pmc_ndt.sh ${TAG}_ndt > /dev/null 2>&1
bash scripts/gpu_pmc_other.sh ${TAG}_other > /dev/null 2>&1
python scripts/pmc_to_json.py gpurun_out/${TAG}_pmc gpurun_out/${TAG}_pmc/pmc_latest.json gpurun_out/${TAG}_ndt gpurun_out/${TAG}_other > /dev/null
cp gpurun_out/${TAG}_pmc/pmc_latest.json profiles/pmc_latest.json   # (so that the plain run below reports traffic + valu-issue)
python bench.py > gpurun_out/${TAG}_bench_line_noprof.json 2> /dev/null
# ... and the driver's own command
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_driver_cmd_bench_line.json 2> /dev/null
ls gpurun_out/${TAG}_* | head -40
