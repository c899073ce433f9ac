out=gpurun_out/ab5.log; : > $out
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
  for lib in tree experiments/ab/libpcgx_head.so; do
    if [ $lib = tree ]; then unset PCGX_LIB; else export PCGX_LIB=$lib; fi
    echo -n "$lib $i: " >> $out
    python bench.py --steps 4000 --warmup 400 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])" >> $out 2>&1
  done
done
python - <<'PY'
import collections
d=collections.defaultdict(list)
for l in open('gpurun_out/ab5.log'):
    a=l.split(); d[a[0]].append(float(a[-1]))
for k,v in d.items(): print(k, "mean %.5f min %.5f" % (sum(v)/len(v), min(v)), " ".join("%.4f"%x for x in v))
PY
