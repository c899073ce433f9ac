#!/bin/bash
# the one-launch Fit's workgroup size (PCGX_SMALL_BLOCK): the reference's ICP benchmark rows with each build
for lib in "" experiments/ab/libpcgx_sb256.so experiments/ab/libpcgx_sb128.so; do
  echo "lib=${lib:-default}"
  PCGX_LIB=$lib PCGX_ICP_SMALL_TARGET=32768 PCGX_ICP_SMALL_BASE=32767 timeout -k 10 300 python tests/perf_rows_ref.py 2>/dev/null | grep ICPGradient | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('  ', d['row'][34:45], d['note'][10:130])
"
done
