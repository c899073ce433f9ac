#!/bin/bash
# Round 6: the ring with the inboxes in device memory (HIP IPC between processes, plain pointers between the slots of one
# process) -- the sharded tests, then REHEARSAL lines of bench.py at 2 / 4 / 5 ranks on the one GPU with the inboxes in
# device memory and, for comparison, in host memory (PCGX_RING_MEM=host: round 5's form).  Five ranks (+ the launcher, which holds the device open too) is what the GPU
# box's process guard allows on its card.
tag=${1:-r6ring}
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_gpu_multi.py tests/test_gpu_sharded_abi.py tests/test_gpu_bench_rehearsal.py "tests/test_gpu_icp.py::test_a_step_without_the_leftover_walk_is_enqueued_again_when_the_grid_leaves_a_target" -x -q -m gpu > gpurun_out/${tag}_tests.log 2>&1
rc=$?
echo tests rc=$rc; tail -15 gpurun_out/${tag}_tests.log
[ $rc -eq 0 ] || exit $rc
for mem in dev host; do
  for n in 2 4 5; do
    PCGX_RING_MEM=$mem PCGX_BENCH_REHEARSE=1 timeout -k 10 200 python bench.py --gpus $n --steps 100 --warmup 20 --points 125000 > gpurun_out/${tag}_${mem}_n${n}.json 2> gpurun_out/${tag}_${mem}_n${n}.err
    echo "mem=$mem n=$n rc=$?"
    python -c "import json; d=json.loads(open('gpurun_out/${tag}_${mem}_n${n}.json').read().strip().splitlines()[-1]); print({k: d.get(k) for k in ('ms_per_step','ms_per_step_min','ms_per_step_max','ms_per_step_f64_tree','shard_stats')})"
  done
done
