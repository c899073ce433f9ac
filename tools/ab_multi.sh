#!/bin/bash
# Same-box comparison of several builds: bench.py with the tree's libpcgx.so and with every experiments/ab/*.so
# (PCGX_LIB), three rounds; then the debug counters of a Fit (runs / failed runs per iteration) for each.
out=gpurun_out/${1:-abm}.log
: > $out
for i in 1 2 3; do
  for lib in tree experiments/ab/*.so; do
    if [ $lib = tree ]; then unset PCGX_LIB; else export PCGX_LIB=$lib; fi
    echo -n "$lib $i: " >> $out
    python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" >> $out 2>&1
  done
done
for lib in tree experiments/ab/*.so; do
  if [ $lib = tree ]; then unset PCGX_LIB; else export PCGX_LIB=$lib; fi
  echo "== $lib" >> $out
  timeout -k 10 200 python tools/strict_probe.py 2>&1 | grep "^iter 1[0-9]\|mismatching\|^strict 1" | cut -c1-90 >> $out
done
cat $out
