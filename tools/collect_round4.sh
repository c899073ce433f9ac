#!/bin/bash
# round-end evidence in one gpurun call: rocprofv3 trace + PMC passes of the bench command (profiles/collect.sh),
# then the bench line and the per-row timings on the same build; the summaries copied under gpurun_out/ for the way
# back (gpurun merges gpurun_out/ only), the raw CSVs dropped (tens of MB)
TAG=${1:-r04a}
mkdir -p gpurun_out
bash profiles/collect.sh $TAG > gpurun_out/${TAG}_collect.log 2>&1
echo collect rc=$?
cp profiles/${TAG}_kernel_stats.csv profiles/${TAG}_pmc.json profiles/${TAG}_fetch_probe.json gpurun_out/ 2>/dev/null
rm -rf gpurun_out/prof_$TAG
timeout -k 10 600 python bench.py > gpurun_out/${TAG}_bench_n1.json 2> gpurun_out/${TAG}_bench_n1.err
echo bench rc=$?
timeout -k 10 900 python tests/perf_rows.py > gpurun_out/${TAG}_rows.json 2> gpurun_out/${TAG}_rows.err
echo rows rc=$?
tail -3 gpurun_out/${TAG}_collect.log
