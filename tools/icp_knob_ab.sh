#!/bin/bash
# Same-box, same-library A/B of one environment knob of the C4 ICP step:  bash tools/icp_knob_ab.sh PCGX_ICP_CERT=0 [tag]
out=gpurun_out/${2:-icpknob}.log
: > $out
for i in 1 2 3; do
  echo "default $(python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])")" >> $out
  echo "$1 $(env $1 python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])")" >> $out
done
cat $out
