#!/bin/bash
# round 3: Range on the grid -- tests, then kernel times (grid and, with PCGX_RANGE_WALK=1, the walk)
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kdtree.py tests/test_gpu_segment.py tests/test_gpu_delete.py -q -m gpu -x > gpurun_out/r3_rangetests.log 2>&1
echo tests rc=$?; tail -3 gpurun_out/r3_rangetests.log
timeout -k 10 200 python tools/range_probe.py 2>&1 | tail -1
bash tools/prof_any.sh rangeprof tools/range_prof.py 2>&1 | grep -i "range\|RangeBatch"
PCGX_RANGE_WALK=1 bash tools/prof_any.sh rangeprof_walk tools/range_prof.py 2>&1 | grep -i "range_kernel\|RangeBatch"
