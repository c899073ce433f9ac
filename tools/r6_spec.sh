#!/bin/bash
# Round 6: the walk ahead of a wait (strict_chain_kernel<., kSpec>) -- the strict tests, C5's 8-chunk share with and
# without it on one GPU, the sharded tests, and the REHEARSAL lines at 2 / 4 ranks with and without it.
tag=${1:-r6spec}
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/test_gpu_strict_rows.py tests/test_gpu_c5.py tests/test_gpu_multi.py tests/test_gpu_sharded_abi.py -x -q -m gpu > gpurun_out/${tag}_tests.log 2>&1
rc=$?
echo tests rc=$rc; tail -8 gpurun_out/${tag}_tests.log
[ $rc -eq 0 ] || exit $rc
show='import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["ms_per_step"], d["ms_per_step_min"], d["ms_per_step_max"], {k: round(v["ms"]*1e3,1) for k,v in d["roofline"]["step"]["kernels"].items()} if "roofline" in d and d["roofline"] else "", d.get("worst_iteration_us"))'
for spec in 1 0; do
  PCGX_STRICT_SPEC=$spec timeout -k 10 500 python bench.py --workload c5 --steps 40 --warmup 20 --no-cpu-baseline > gpurun_out/${tag}_c5_spec${spec}.json 2> gpurun_out/${tag}_c5_spec${spec}.err
  echo "c5 share, spec=$spec rc=$?"; python -c "$show" gpurun_out/${tag}_c5_spec${spec}.json
  for n in 2 4; do
    PCGX_STRICT_SPEC=$spec PCGX_BENCH_REHEARSE=1 timeout -k 10 200 python bench.py --gpus $n --steps 100 --warmup 20 --points 125000 > gpurun_out/${tag}_spec${spec}_n${n}.json 2> gpurun_out/${tag}_spec${spec}_n${n}.err
    echo "rehearsal spec=$spec n=$n rc=$?"; python -c "$show" gpurun_out/${tag}_spec${spec}_n${n}.json
  done
done
timeout -k 10 200 python tools/strict_probe.py 8000000 2>&1 | grep -v amdgpu.ids | tail -60 > gpurun_out/${tag}_probe8m.txt; tail -12 gpurun_out/${tag}_probe8m.txt
