#!/bin/bash
# the chunk-parallel chain kernel on one GPU: the strict tests, the C4 bench line, the C5 share's bench line
#   gpurun -- 'bash tools/r5_chain.sh [tag] [quick]'
tag=${1:-r5chain}
mkdir -p gpurun_out
if [ -z "$2" ]; then
timeout -k 10 900 python -m pytest tests/test_gpu_strict_rows.py tests/test_gpu_icp.py tests/test_gpu_c5.py -x -q -m gpu > gpurun_out/${tag}_tests.log 2>&1
echo tests rc=$?; tail -3 gpurun_out/${tag}_tests.log
fi
show='import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["ms_per_step"], d["ms_per_step_min"], d["ms_per_step_max"], d["value"], {k: round(v["ms"]*1e3,1) for k,v in d["roofline"]["step"]["kernels"].items()}, d["worst_iteration_us"])'
timeout -k 10 300 python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-extras > gpurun_out/${tag}_bench_c4.json 2> gpurun_out/${tag}_bench_c4.err
echo c4 rc=$?; python -c "$show" gpurun_out/${tag}_bench_c4.json
PCGX_STRICT_REPAIR=0 timeout -k 10 300 python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-extras > gpurun_out/${tag}_bench_c4_norepair.json 2> gpurun_out/${tag}_bench_c4_norepair.err
echo c4 without the repair pass rc=$?; python -c "$show" gpurun_out/${tag}_bench_c4_norepair.json
timeout -k 10 600 python bench.py --workload c5 --steps 40 --warmup 20 > gpurun_out/${tag}_bench_c5.json 2> gpurun_out/${tag}_bench_c5.err
echo c5 rc=$?; python -c "$show" gpurun_out/${tag}_bench_c5.json
