#!/bin/bash
mkdir -p gpurun_out
rm -f gpurun_out/r3_trace_std.txt
PCGX_STRICT_TRACE=gpurun_out/r3_trace_std.txt timeout -k 10 300 python tools/strict_probe.py > gpurun_out/r3_trace_std.log 2>&1
echo rc=$?
python3 - <<'PY'
import numpy as np
blocks = open("gpurun_out/r3_trace_std.txt").read().split("#\n")
for bi in (2, 8):
    rows = [list(map(int, l.split())) for l in blocks[bi].strip().split("\n")]
    a = np.array(rows, dtype=np.int64)
    sum_end = a[:, 5].max()
    j = a[a[:, 6] > 0]
    j = j[j[:, 6] >= sum_end - 100000]  # stamps of this launch only (stale ones are older)
    base = j[:, 6].min()
    print("launch %d: summary kernel's last workgroup ends at 0; job workgroups with stamps: %d" % (bi, len(j)))
    print("   first job workgroup enters %.1f us after that; entries spread over %.1f us" % ((base - sum_end) / 100.0, (j[:, 6].max() - base) / 100.0))
    for what, name in ((1, "crossing cand"), (2, "no window cand"), (3, "first tile"), (4, "plain stands"), (5, "crossing scan"), (6, "no window scan")):
        k = j[(j[:, 10] & 0xff) == what]
        if len(k) == 0: continue
        ends = np.maximum(k[:, 8], k[:, 9])
        print("   %-12s %3d: enter %.1f..%.1f us, loads %.1f us, ends %.1f..%.1f us after the first entry (own duration %.1f..%.1f)" % (
            name, len(k), (k[:, 6].min() - base) / 100.0, (k[:, 6].max() - base) / 100.0, np.mean(k[:, 7] - k[:, 6]) / 100.0,
            (ends.min() - base) / 100.0, (ends.max() - base) / 100.0, (ends - k[:, 6]).min() / 100.0, (ends - k[:, 6]).max() / 100.0))
        if what == 1:
            print("       scans (waves 0 / 1) end at %.1f..%.1f us after the first entry" % ((k[:, 13].min() - base) / 100.0, (k[:, 13].max() - base) / 100.0))
PY
