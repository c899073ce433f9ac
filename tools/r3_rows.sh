#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_voxel.py -q -m gpu -k "c3 or golden" > gpurun_out/r3_seamtests.log 2>&1
echo tests rc=$?; tail -2 gpurun_out/r3_seamtests.log
timeout -k 10 900 python tests/perf_rows.py > gpurun_out/r3_rows.json 2> gpurun_out/r3_rows.err
echo rows rc=$?; tail -2 gpurun_out/r3_rows.err
