#!/bin/bash
mkdir -p gpurun_out
rm -f gpurun_out/r3_trace_std.txt
PCGX_STRICT_TRACE=gpurun_out/r3_trace_std.txt timeout -k 10 300 python tools/strict_probe.py > gpurun_out/r3_trace_std.log 2>&1
echo rc=$?; grep "^strict 1\|final\|MISMATCH" gpurun_out/r3_trace_std.log
bash tools/prof_any.sh r3a_strict_std tools/strict_prof.py > gpurun_out/r3a_prof_std.txt 2>&1; sed -n 3,9p gpurun_out/r3a_prof_std.txt
timeout -k 10 300 python tools/strict_hover_probe.py > gpurun_out/r3_hover_probe.log 2>&1
echo "hover rc=$?"; grep "^strict" gpurun_out/r3_hover_probe.log
