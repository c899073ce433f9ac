#!/bin/bash
# kNN C2: one query per lane against four lanes per query (PCGX_KNN_COOP=1): parity tests, then timings
tag=${1:-r5knn}
mkdir -p gpurun_out
PCGX_KNN_COOP=1 timeout -k 10 600 python -m pytest tests/test_gpu_grid.py tests/test_gpu_kdtree.py -x -q -m gpu > gpurun_out/${tag}_tests_coop.log 2>&1
echo "coop tests rc=$?"; tail -2 gpurun_out/${tag}_tests_coop.log
for i in 1 2 3; do
  echo "lane  $(python tools/knn_time.py 2>/dev/null | tail -3 | tr '\n' ' ')"
  echo "coop  $(PCGX_KNN_COOP=1 python tools/knn_time.py 2>/dev/null | tail -3 | tr '\n' ' ')"
done
