#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 120 python tools/strict_probe.py 20000 > gpurun_out/r3_strict_probe_small.log 2>&1
echo small rc=$?; grep "^strict 1\|final\|MISMATCH" gpurun_out/r3_strict_probe_small.log | head -5
timeout -k 10 300 python tools/strict_probe.py > gpurun_out/r3_strict_probe.log 2>&1
echo rc=$?; grep "^strict 1\|final\|MISMATCH" gpurun_out/r3_strict_probe.log | head
timeout -k 10 300 python tools/strict_hover_probe.py > gpurun_out/r3_hover_probe.log 2>&1
echo "hover rc=$?"; grep "^strict" gpurun_out/r3_hover_probe.log
bash tools/prof_any.sh r3a_strict_std tools/strict_prof.py > gpurun_out/r3a_prof_std.txt 2>&1; grep "strict_\|icp_" gpurun_out/r3a_prof_std.txt | head -8
