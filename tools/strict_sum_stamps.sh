#!/bin/bash
# per-workgroup wall-clock stamps of strict_sum_kernel (PCGX_STRICT_TRACE): start, terms formed, exchange done, end
mkdir -p gpurun_out
rm -f gpurun_out/r4_trace.txt
PCGX_STRICT_TRACE=gpurun_out/r4_trace.txt timeout -k 10 300 python tools/strict_trace_run.py > gpurun_out/r4_trace.log 2>&1
echo rc=$?; grep "^strict 1\|final\|MISMATCH" gpurun_out/r4_trace.log
python3 - <<'PY'
import numpy as np
blocks = open("gpurun_out/r4_trace.txt").read().split("#\n")
for bi in (1, 5, 10):
    rows = [list(map(int, l.split())) for l in blocks[bi].strip().split("\n")]
    a = np.array(rows, dtype=np.int64)
    t0, t1, tx, t5 = a[:, 0], a[:, 1], a[:, 2], a[:, 5]
    base = t0.min()
    us = lambda v: v / 100.0
    print("launch %d: %d workgroups; last start %.1f us, last end %.1f us" % (bi, len(a), us(t0.max() - base), us(t5.max() - base)))
    if tx.max() > 0:
        print("   terms formed at %.1f (median) .. %.1f (last) us; exchange done at %.1f (first) %.1f (median) %.1f (last) us; phase 2 %.2f us (p90 %.2f)" % (
            us(np.median(tx) - base), us(tx.max() - base), us(t1.min() - base), us(np.median(t1) - base), us(t1.max() - base),
            us(np.mean(t5 - t1)), us(np.percentile(t5 - t1, 90))))
        k = np.argsort(tx)
        print("   by tile index (every 61st): ", " ".join("%d:%.1f/%.1f" % (i, us(tx[i] - base), us(t1[i] - base)) for i in range(0, len(a), 61)))
    else:
        print("   phase 1 %.2f us (p90 %.2f), phase 2 %.2f us (p90 %.2f)" % (us(np.mean(t1 - t0)), us(np.percentile(t1 - t0, 90)), us(np.mean(t5 - t1)), us(np.percentile(t5 - t1, 90))))
PY
