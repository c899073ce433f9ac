#!/bin/bash
# Sweep the walk kernel's launch knobs on the ICP step (run on the GPU box): one bench line per setting.
out=gpurun_out/sweep.log
: > $out
for b in 2 3 4; do for r in 4 8 16 24; do
  echo "blocks=$b refill=$r" >> $out
  PCGX_WALK_BLOCKS_PER_CU=$b PCGX_WALK_REFILL=$r timeout 120 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'])" >> $out
done; done
