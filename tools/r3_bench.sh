#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 500 python bench.py > gpurun_out/r3_bench.json 2> gpurun_out/r3_bench.err
echo bench rc=$?; tail -3 gpurun_out/r3_bench.err
timeout -k 10 500 python -m pytest tests/test_gpu_bench_rehearsal.py -q -m gpu > gpurun_out/r3_rehearsal.log 2>&1
echo rehearsal rc=$?; tail -5 gpurun_out/r3_rehearsal.log
