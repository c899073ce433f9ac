#!/bin/bash
# the ring form of the sharded reference sums on one GPU: slots of one process, processes over shared memory
tag=${1:-r5ring}
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/test_gpu_multi.py tests/test_gpu_sharded_abi.py -x -q -m gpu > gpurun_out/${tag}_tests.log 2>&1
echo tests rc=$?; tail -15 gpurun_out/${tag}_tests.log
