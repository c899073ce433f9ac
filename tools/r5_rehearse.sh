#!/bin/bash
# REHEARSAL of the N > 1 bench line on the one-GPU box (PCGX_BENCH_REHEARSE=1: every rank on cuda:0, callback
# communicator over gloo, the ring in shared memory): 2 and 4 ranks (a box allows six processes on its GPU), at
# 125k points per rank and at C4's 1M; not a measurement of scaling -- the ranks share one GPU
tag=${1:-r5rehearse}
mkdir -p gpurun_out
python -m pytest tests/test_gpu_bench_rehearsal.py -x -q -m gpu > gpurun_out/${tag}_tests.log 2>&1
echo tests rc=$?; tail -3 gpurun_out/${tag}_tests.log
for n in 2 4; do
  for pts in 125000 1000000; do
    PCGX_BENCH_REHEARSE=1 timeout -k 10 300 python bench.py --gpus $n --steps 100 --warmup 20 --points $pts > gpurun_out/${tag}_n${n}_p${pts}.json 2> gpurun_out/${tag}_n${n}_p${pts}.err
    echo "n=$n pts=$pts rc=$?"
    python -c "import json; d=json.loads(open('gpurun_out/${tag}_n${n}_p${pts}.json').read().strip().splitlines()[-1]); print({k: d.get(k) for k in ('ms_per_step','ms_per_step_min','ms_per_step_max','value','ms_per_step_f64_tree','shard_stats')})"
  done
done
