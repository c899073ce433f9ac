#!/bin/bash
# Same-box, same-library A/B of one environment knob of the kNN C2 call:  bash tools/knn_knob_ab.sh PCGX_KNN_REFINE=0 [tag]
out=gpurun_out/${2:-knnknob}.log
: > $out
for i in 1 2 3; do
  echo "default $(python tools/knn_time.py 2>/dev/null | tail -2 | tr '\n' ' ')" >> $out
  echo "$1 $(env $1 python tools/knn_time.py 2>/dev/null | tail -2 | tr '\n' ' ')" >> $out
done
cat $out
