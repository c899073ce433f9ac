#!/bin/bash
# the chunk-parallel chain kernel on one GPU: the strict tests, the C4 bench line, the C5 share's bench line
#   gpurun -- 'bash tools/r5_chain.sh [tag]'
tag=${1:-r5chain}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_strict_rows.py tests/test_gpu_icp.py -x -q -m gpu > gpurun_out/${tag}_tests.log 2>&1
echo tests rc=$?; tail -3 gpurun_out/${tag}_tests.log
timeout -k 10 300 python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-extras > gpurun_out/${tag}_bench_c4.json 2> gpurun_out/${tag}_bench_c4.err
echo c4 rc=$?; python -c "import json,sys; d=json.loads(open('gpurun_out/${tag}_bench_c4.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], {k: round(v['ms']*1e3,1) for k,v in d['roofline']['step']['kernels'].items()})"
timeout -k 10 600 python bench.py --workload c5 --steps 40 --warmup 20 > gpurun_out/${tag}_bench_c5.json 2> gpurun_out/${tag}_bench_c5.err
echo c5 rc=$?; python -c "import json,sys; d=json.loads(open('gpurun_out/${tag}_bench_c5.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], {k: round(v['ms']*1e3,1) for k,v in d['roofline']['step']['kernels'].items()})"
