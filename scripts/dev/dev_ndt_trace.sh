#!/bin/bash
# kernel trace of NDT registrations of the 2M ring pair: what lies between two derivative passes
out=$PWD/gpurun_out/${1:-ndt_trace}
mkdir -p $out
export TMPDIR=/tmp
NDT_BLOCKS=0 rocprofv3 --kernel-trace -d $out/prof -o p -- python scripts/dev/dev_ndt_blocks.py > $out/run.log 2>&1
ls $out/prof
