#!/bin/bash
# round 3: strict-sum pipeline check + timing (one gpurun call)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_icp.py -x -q -m gpu > gpurun_out/r3_icp_tests.log 2>&1
echo "icp tests rc=$?" | tee -a gpurun_out/r3_icp_tests.log
tail -5 gpurun_out/r3_icp_tests.log
timeout -k 10 300 python tools/strict_probe.py > gpurun_out/r3_strict_probe.log 2>&1
echo "probe rc=$?"; tail -4 gpurun_out/r3_strict_probe.log
timeout -k 10 300 python tools/strict_hover_probe.py > gpurun_out/r3_hover_probe.log 2>&1
echo "hover rc=$?"; grep "^strict" gpurun_out/r3_hover_probe.log
