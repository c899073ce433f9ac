#!/bin/bash
# Same-box comparison of the C3 filter call over library variants (tools/mk_variant.sh):  bash tools/voxel_variants.sh tag name...
tag=$1; shift
out=gpurun_out/$tag.log
: > $out
for i in 1 2 3; do
  echo "tree $(python tools/voxel_probe.py 2>/dev/null | grep 'voxel ms')" >> $out
  for v in "$@"; do
    echo "$v $(PCGX_LIB=experiments/ab/libpcgx_$v.so python tools/voxel_probe.py 2>/dev/null | grep 'voxel ms')" >> $out
  done
done
cat $out
