#!/bin/bash
# round-end artifacts: bench line (N = 1), per-row timings, kernel stats + PMC summary
TAG=${1:-r03a}
mkdir -p gpurun_out
timeout -k 10 600 python bench.py > gpurun_out/${TAG}_bench_n1.json 2> gpurun_out/${TAG}_bench_n1.err
echo bench rc=$?
timeout -k 10 900 python tests/perf_rows.py > gpurun_out/${TAG}_rows.json 2> gpurun_out/${TAG}_rows.err
echo rows rc=$?
