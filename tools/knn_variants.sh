#!/bin/bash
# Same-box comparison of the kNN C2 call over library variants (tools/mk_variant.sh):  bash tools/knn_variants.sh tag name...
tag=$1; shift
out=gpurun_out/$tag.log
: > $out
for i in 1 2 3; do
  echo "tree $(python tools/knn_time.py 2>/dev/null | tail -2 | tr '\n' ' ')" >> $out
  for v in "$@"; do
    echo "$v $(PCGX_LIB=experiments/ab/libpcgx_$v.so python tools/knn_time.py 2>/dev/null | tail -2 | tr '\n' ' ')" >> $out
  done
done
cat $out
