#!/bin/bash
# Developer (run HERE, then gpurun the printed command): builds wm_gicp.o with -DWM_COV_COUNT -DWM_COV_BLOCK=$1
B=$1; EXTRA=$2
cd "$(dirname "$0")/../../libwave_amd/csrc" || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-result $EXTRA -DWM_COV_BLOCK=$B -c wm_gicp.hip -o wm_gicp.o 2>&1 | grep -i "error" -A3
make -s 2>&1 | grep -i error
