#!/bin/bash
mkdir -p gpurun_out
for m in 0 4 16 32; do
PCGX_STRICT_POLL=$m timeout -k 10 300 python tools/strict_probe.py > gpurun_out/r3_poll_$m.log 2>&1
echo "poll mode $m rc=$?"; grep "waits\|^strict 1\|MISMATCH" gpurun_out/r3_poll_$m.log
done
