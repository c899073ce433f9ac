#!/bin/bash
# Same-box A/B of two builds of the library: bench.py alternates between the tree's libpcgx.so and
# experiments/ab/libpcgx_head.so (PCGX_LIB), three times each; then the strict tests on the tree's build.
#   gpurun -- 'bash tools/ab.sh [tag]'
tag=${1:-ab}
out=gpurun_out/${tag}.log
: > $out
for i in 1 2 3; do
  echo "== new $i" >> $out
  python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" >> $out 2>&1
  echo "== old $i" >> $out
  PCGX_LIB=experiments/ab/libpcgx_head.so python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" >> $out 2>&1
done
python -m pytest tests/test_gpu_strict_rows.py tests/test_gpu_icp.py tests/test_gpu_multi.py -x -q -m gpu 2>&1 | tail -3 >> $out
cat $out
