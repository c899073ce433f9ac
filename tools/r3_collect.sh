#!/bin/bash
# round-end evidence in one gpurun call: rocprofv3 trace + PMC passes of the bench command (profiles/collect.sh),
# then the bench line and the per-row timings on the same build; everything copied under gpurun_out/ for the way back
TAG=${1:-r03b}
mkdir -p gpurun_out
bash profiles/collect.sh $TAG > gpurun_out/${TAG}_collect.log 2>&1
echo collect rc=$?
cp profiles/${TAG}_kernel_stats.csv profiles/${TAG}_pmc.json profiles/${TAG}_fetch_probe.json gpurun_out/ 2>/dev/null
rm -rf gpurun_out/prof_$TAG/pmc_* gpurun_out/prof_$TAG/probe_* gpurun_out/prof_$TAG/trace   # (raw CSVs: tens of MB)
bash tools/r3_final.sh $TAG
