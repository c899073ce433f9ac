#!/bin/bash
# per-workgroup wall-clock stamps of strict_sum_kernel (PCGX_STRICT_TRACE): start, terms formed, end
mkdir -p gpurun_out
rm -f gpurun_out/r3_trace_std.txt
PCGX_STRICT_TRACE=gpurun_out/r3_trace_std.txt timeout -k 10 300 python tools/strict_probe.py > gpurun_out/r3_trace_std.log 2>&1
echo rc=$?; grep "^strict 1\|final\|MISMATCH" gpurun_out/r3_trace_std.log
python3 - <<'PY'
import numpy as np
blocks = open("gpurun_out/r3_trace_std.txt").read().split("#\n")
for bi in (1, 5, 10):
    rows = [list(map(int, l.split())) for l in blocks[bi].strip().split("\n")]
    a = np.array(rows, dtype=np.int64)
    t0, t1, t5 = a[:, 0], a[:, 1], a[:, 5]
    base = t0.min()
    print("launch %d: %d workgroups; first start 0, last start %.1f us, last end %.1f us; phase 1 %.2f us (p90 %.2f), phase 2 %.2f us (p90 %.2f), workgroup %.2f us" % (
        bi, len(a), (t0.max() - base) / 100.0, (t5.max() - base) / 100.0, np.mean(t1 - t0) / 100.0, np.percentile(t1 - t0, 90) / 100.0,
        np.mean(t5 - t1) / 100.0, np.percentile(t5 - t1, 90) / 100.0, np.mean(t5 - t0) / 100.0))
    order = np.argsort(t0)
    # concurrency: workgroups running at the middle of the launch
    mid = base + (t5.max() - base) // 2
    print("   running at mid-launch: %d" % int(((t0 <= mid) & (t5 > mid)).sum()))
PY
