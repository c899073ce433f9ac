#!/bin/bash
# Memory-pipeline counters for the walk kernel (separate --pmc passes; kernel-trace only).
set -u
TAG=${1:-r00}
OUT=gpurun_out/pmcmem_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
BENCH="python3 bench.py --steps 40 --warmup 20 --no-cpu-baseline --no-extras"
i=0
# at most two counters of a block per pass; every pass under its own timeout (a rejected
# counter set makes rocprofv3 abort and then hang)
for PASS in "TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
            "TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum" \
            "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum" \
            "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES" "GRBM_GUI_ACTIVE GRBM_TA_BUSY"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $PASS --output-format csv -d $OUT/pmc_$i -- $BENCH > $OUT/pmc_$i.log 2>&1
done
python3 profiles/summarize.py $OUT mem_$TAG
