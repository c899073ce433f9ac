#!/bin/bash
# Same-box A/B of the voxel filter (C3, device resident): the tree's library against experiments/ab/libpcgx_head.so
out=gpurun_out/${1:-voxab}.log
: > $out
for i in 1 2 3 4; do
  echo "new $(python tools/voxel_probe.py 2>/dev/null | grep 'voxel ms')" >> $out
  echo "old $(PCGX_LIB=experiments/ab/libpcgx_head.so python tools/voxel_probe.py 2>/dev/null | grep 'voxel ms')" >> $out
done
cat $out
