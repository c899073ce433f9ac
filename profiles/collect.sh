#!/bin/bash
# Collects the rocprofv3 evidence for one round on the GPU box (run via gpurun from the repo root):
#   bash profiles/collect.sh r02
# 1) kernel trace + stats of the exact bench.py command; 2) PMC passes (separate runs, no tracing
#    domains besides kernel-trace) for HBM traffic and issue/occupancy counters, WITH the extras
#    (kNN C2, VoxelGrid C3, f64-tree ICP) so that their kernels have FETCH/WRITE rows too;
# 3) the same two byte counters on access patterns with a known byte count (tools/fetch_probe.py).
set -u
TAG=${1:-r00}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
BENCH="python3 bench.py --steps 40 --warmup 20 --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/bench_trace.log 2>&1
for PASS in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
            "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU" \
            "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"; do
  NAME=$(echo $PASS | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $PASS --output-format csv -d $OUT/pmc_$NAME -- $BENCH > $OUT/pmc_$NAME.log 2>&1
done
for PASS in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout 200 rocprofv3 --kernel-trace --pmc $PASS --output-format csv -d $OUT/probe_$PASS -- python3 tools/fetch_probe.py > $OUT/probe_$PASS.log 2>&1
done
python3 profiles/summarize.py $OUT $TAG
