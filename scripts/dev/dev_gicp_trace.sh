#!/bin/bash
# kernel trace of GICP 500k registrations (statistics objective): the timeline of one registration
out=$PWD/gpurun_out/${1:-gicp_trace}
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $out/prof -o p -- python scripts/dev/dev_gicp_quad_time.py > $out/run.log 2>&1
ls $out/prof
