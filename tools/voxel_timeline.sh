#!/bin/bash
# Where a C3 filter call's time goes on the GPU: every kernel of the steady-state calls with its start (relative to the
# call's first kernel), duration and the gap in front of it -- median over the calls of tools/voxel_probe.py.
#   bash tools/voxel_timeline.sh [tag]          (PROBE_N=<points> for another cloud size)
TAG=${1:-voxtl}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 tools/voxel_probe.py $PROBE_N > $OUT/probe.log 2>&1
tail -1 $OUT/probe.log
python3 - "$OUT" <<'PY' | tee $OUT/timeline.txt
import csv, glob, sys, statistics
rows = []
for p in glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("<")[0][-30:]))
rows.sort()
# a call ends with its result words on their way to the host (a copy command, or the one-wave kernel)
calls, cur = [], []
for s, e, name in rows:
    cur.append((s, e, name))
    if not any(w in name for w in ("pcgx", "minmax", "vb_", "copyBuffer", "radix", "voxel")):
        cur.pop()  # (the probe's own kernels between calls)
    if "copyBuffer" in name or "read_back_kernel" in name:
        calls.append(cur); cur = []
calls = [c for c in calls if any("bucket_kernel" in n for _, _, n in c)][5:]
k = min(len(c) for c in calls)
calls = [c for c in calls if len(c) == k]
print("calls", len(calls), "kernels per call", k)
tot = []
for i in range(k):
    st = statistics.median((c[i][0] - c[0][0]) / 1e3 for c in calls)
    du = statistics.median((c[i][1] - c[i][0]) / 1e3 for c in calls)
    gap = statistics.median(((c[i][0] - c[i - 1][1]) / 1e3 if i else 0.0) for c in calls)
    print("%-32s start %8.1f  dur %8.1f  gap before %6.1f" % (calls[0][i][2], st, du, gap))
print("first kernel start -> last kernel end: %.1f us" % statistics.median((c[-1][1] - c[0][0]) / 1e3 for c in calls))
print("call to call: %.1f us" % statistics.median((calls[j + 1][0][0] - calls[j][0][0]) / 1e3 for j in range(len(calls) - 1)))
PY
rm -rf $OUT/trace
