#!/bin/bash
show='import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(round(d["ms_per_step"],5), {k: round(v["ms"]*1e3,1) for k,v in d["roofline"]["step"]["kernels"].items() if "chain" in k})'
for lib in "" experiments/ab/libpcgx_spec2.so experiments/ab/libpcgx_spec1.so; do
  for depth in 1 2 4; do
    PCGX_LIB=$lib PCGX_STRICT_SPEC_DEPTH=$depth timeout -k 10 300 python bench.py --workload c5 --steps 40 --warmup 20 --no-cpu-baseline > gpurun_out/spec_ab.json 2>/dev/null
    echo "lib=${lib:-per4} depth=$depth c5: $(python -c "$show" gpurun_out/spec_ab.json)"
  done
done
for lib in "" experiments/ab/libpcgx_spec2.so experiments/ab/libpcgx_spec1.so; do
  PCGX_LIB=$lib timeout -k 10 300 python -m pytest "tests/test_gpu_c5.py::test_c5_walks_ahead_of_the_chunks_hand_overs" -x -q -m gpu 2>&1 | tail -2
done
