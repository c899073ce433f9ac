#!/bin/bash
# Same-box, same-library A/B of one environment knob of the voxel filter (C3, device resident):
#   bash tools/voxel_knob_ab.sh PCGX_VOXEL_BUCKET_BOUNDS_KERNEL=1 [tag]
out=gpurun_out/${2:-voxknob}.log
: > $out
for i in 1 2 3 4 5 6; do
  echo "default $(python tools/voxel_probe.py 2>/dev/null | grep 'voxel ms')" >> $out
  echo "$1 $(env $1 python tools/voxel_probe.py 2>/dev/null | grep 'voxel ms')" >> $out
done
cat $out
