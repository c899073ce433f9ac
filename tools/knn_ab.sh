#!/bin/bash
# Same-box A/B of the kNN C2 call: the tree's library against experiments/ab/libpcgx_head.so
out=gpurun_out/${1:-knnab}.log
: > $out
for i in 1 2 3; do
  echo "new $(python tools/knn_time.py 2>/dev/null | tail -2 | tr '\n' ' ')" >> $out
  echo "old $(PCGX_LIB=experiments/ab/libpcgx_head.so python tools/knn_time.py 2>/dev/null | tail -2 | tr '\n' ' ')" >> $out
done
cat $out
