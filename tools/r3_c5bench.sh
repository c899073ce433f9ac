#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python bench.py --workload c5 --steps 40 --warmup 20 > gpurun_out/r3_bench_c5.json 2> gpurun_out/r3_bench_c5.err
echo c5 bench rc=$?; tail -3 gpurun_out/r3_bench_c5.err; cut -c1-600 gpurun_out/r3_bench_c5.json
MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout -k 10 300 python tools/rccl_two_rank_probe.py > gpurun_out/r3_rccl_probe1.log 2>&1
echo probe rc=$?; tail -2 gpurun_out/r3_rccl_probe1.log
