#!/bin/bash
# Issue / wait breakdown of the ICP correspondence kernel (separate --pmc passes; kernel-trace only).
# SQ_WAIT_ANY + SQ_WAIT_INST_ANY + SQ_ACTIVE_INST_ANY ~= SQ_WAVE_CYCLES (quad-cycles, per wave summed).
set -u
TAG=${1:-r00}
OUT=gpurun_out/pmcsq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
BENCH="python3 bench.py --steps 40 --warmup 20 --no-cpu-baseline --no-extras"
i=0
for PASS in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
            "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" \
            "SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" \
            "TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $PASS --output-format csv -d $OUT/pmc_$i -- $BENCH > $OUT/pmc_$i.log 2>&1
  echo "pass $i rc=$?" >> $OUT/passes.log
done
python3 profiles/summarize.py $OUT sq_$TAG
