#!/bin/bash
# points-per-cell sweep of the grid (GPU box):  bash tools/occ_sweep.sh [occ ...]
for occ in ${@:-2 3 4 6}; do
  echo "== occ $occ"
  PCGX_GRID_OCC=$occ timeout -k 10 200 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; e=d['extra']
print(round(d['value']), round(d['ms_per_step']*1e3,1), round(r['kernel_ms']*1e3,1), 'pts', round(r['point_records_per_target'],2), '| knn', round(e['knn_c2_presort']['ms_per_call']*1e3,1), round(e['knn_c2_presort']['grid_kernel_ms']*1e3,1), round(e['knn_c2_unsorted']['ms_per_call']*1e3,1), e['knn_c2_presort']['queries_left_to_walk'], round(e['knn_c2_presort']['point_records_per_query'],1))"
done
