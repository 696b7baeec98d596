#!/bin/bash
# kernel trace of the one-rank sharded registration (rocpd database; read it with scripts/dev/dev_trace_db.py)
tag=${1:-shard1}
out=$PWD/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
B="bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --no-in-flight"
WM_BENCH_FORCE_SHARDED=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29512 \
  rocprofv3 --kernel-trace -d $out/sharded -o s -- python $B > $out/sharded.log 2>&1
if [ "$2" = "plain" ]; then rocprofv3 --kernel-trace -d $out/plain -o p -- python $B > $out/plain.log 2>&1; fi
ls -la $out/sharded
