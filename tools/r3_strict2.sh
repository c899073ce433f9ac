#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python tools/strict_probe.py > gpurun_out/r3_strict_probe.log 2>&1
echo "probe rc=$?"; grep "sum kernel\|^strict\|MISMATCH\|final" gpurun_out/r3_strict_probe.log
bash tools/prof_any.sh r3a_strict tools/strict_prof.py | head -8
