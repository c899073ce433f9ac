#!/bin/bash
# min/max pass of the voxel filter at C3 (10M points): blocks x groups in flight
mkdir -p gpurun_out
for cfg in "1024 2" "2048 2" "4096 2" "1024 4" "2048 4" "4096 4" "2048 1" "8192 1"; do
  set -- $cfg
  PCGX_MM_BLOCKS=$1 PCGX_MM_U=$2 bash tools/voxel_prof.sh mm_$1_$2 2>&1 | grep -E "voxel ms|minmax" | tr '\n' ' '
  echo " <- blocks $1, in flight $2"
done
