#!/bin/bash
# round 3: every ICP-side GPU test (strict sums, sharded ABI, plane, C5, delete) + the strict probes
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_icp.py tests/test_gpu_c5.py tests/test_gpu_icp_plane.py tests/test_gpu_sharded_abi.py tests/test_gpu_delete.py tests/test_gpu_grid.py tests/test_gpu_bench_rehearsal.py -x -q -m gpu > gpurun_out/r3_icp_all.log 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/r3_icp_all.log
timeout -k 10 300 python tools/strict_probe.py > gpurun_out/r3_strict_probe.log 2>&1
echo "probe rc=$?"; grep "^strict\|mismatch" gpurun_out/r3_strict_probe.log
