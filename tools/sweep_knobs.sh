#!/bin/bash
# Sweep walk-kernel knobs on the ICP step and the C2 kNN batch (run on the GPU box).
out=gpurun_out/sweep.log
: > $out
for t in 0 4; do for c in 1 2; do for r in 8 16 24 32; do
  echo -n "tight=$t chunks=$c refill=$r : " >> $out
  PCGX_WALK_TIGHT=$t PCGX_WALK_CHUNKS_PER_REFILL=$c PCGX_WALK_REFILL=$r timeout 120 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), 'knn', round(d['extra']['knn_c2_presort']['walk_kernel_ms'],4), 'plane', round(d['extra']['icp_plane_c4']['corr_kernel_ms'],4))" >> $out
done; done; done
cat $out
