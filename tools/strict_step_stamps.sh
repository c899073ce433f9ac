#!/bin/bash
# timeline of one strict step (PCGX_STRICT_TRACE): summary kernel's end, job workgroups, the rows' walks
mkdir -p gpurun_out
rm -f gpurun_out/r4_ftrace.txt
PCGX_STRICT_TRACE=gpurun_out/r4_ftrace.txt timeout -k 10 300 python tools/strict_trace_run.py > gpurun_out/r4_ftrace.log 2>&1
echo rc=$?
python3 - <<'PY'
import numpy as np
blocks = open("gpurun_out/r4_ftrace.txt").read().split("#\n")
for bi in (2, 8, 12):
    rows = [list(map(int, l.split())) for l in blocks[bi].strip().split("\n")]
    a = np.array(rows, dtype=np.int64)
    sum_end = a[:, 5].max()
    us = lambda v: (v - sum_end) / 100.0
    j = a[a[:, 6] > 0]
    j = j[j[:, 6] >= sum_end - 3000]
    print("launch %d: summary kernel's last workgroup ends at 0 (its first started at %.1f); job workgroups with stamps: %d" % (bi, us(a[:, 0].min()), len(j)))
    for what, name in ((1, "crossing cand"), (2, "no window cand"), (3, "first tile"), (4, "plain stands"), (5, "crossing scan"), (6, "no window scan")):
        k = j[(j[:, 10] & 0xff) == what]
        if len(k) == 0: continue
        ends = np.maximum(k[:, 8], k[:, 9])
        print("   %-14s %3d: enter %.1f..%.1f us, loads %.1f us, ends %.1f..%.1f us" % (
            name, len(k), us(k[:, 6].min()), us(k[:, 6].max()), np.mean(k[:, 7] - k[:, 6]) / 100.0, us(ends.min()), us(ends.max())))
    k = j[((j[:, 10] & 0xff) == 1)]
    k = k[(k[:, 11] > k[:, 6]) & (k[:, 14] > k[:, 6]) & (k[:, 14] < k[:, 6] + 3000)]
    if len(k):
        print("   level crossings (%d), wave 0 after the terms are in LDS: guesses %.1f, own piece %.1f, all pieces %.1f, scan %.1f; tables %.1f us (means)" % ((len(k),) + tuple(
            np.mean(k[:, c] - k[:, 7]) / 100.0 for c in (11, 12, 13, 14, 9))))
    f0 = a[0]
    if f0[15] > f0[11] > 0:
        print("   first tile of row 2: guesses at %.1f, own piece %.1f, all pieces %.1f, scan %.1f, walked %.1f us (after the summary kernel's end)" % tuple(us(f0[c]) for c in (11, 12, 13, 14, 15)))
    print("   the last row's ticket: sums read at %.1f, pose updated at %.1f us" % (us(a[9, 14]), us(a[9, 15])))
    for r in range(9):
        if a[r, 11] == 0: continue
        print("   row %d: enters %.1f, records in %.1f, walk starts %.1f, ends %.1f us; runs %d (failed %d, records %d, tables %d)" % (
            r, us(a[r, 11]), us(a[r, 12]), us(a[r, 13]), us(a[r, 15]), a[r, 14] & 0xffff, (a[r, 14] >> 16) & 0xffff, (a[r, 14] >> 32) & 0xffff, (a[r, 14] >> 48) & 0xffff))
PY
