#!/bin/bash
# Four host-pointer Fits from four threads under rocprofv3 --kernel-trace: per kernel its mean duration while ONE Fit
# runs and while FOUR do, and how many launches of a kind are in flight at once (concurrency).
TAG=${1:-conc4}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 tools/conc4_probe.py 4 > $OUT/run.log 2>&1
tail -2 $OUT/run.log
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
rows = []
for p in glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("<")[0][-28:], r.get("Stream_Id", r.get("Queue_Id", "?"))))
rows.sort()
t_end = rows[-1][1]
# the concurrent phase: the last stretch in which more than one stream is active
streams = collections.Counter(r[3] for r in rows)
print("launches per stream/queue:", dict(streams))
by = collections.defaultdict(list)
for s, e, n, q in rows:
    by[n].append((s, e, q))
def overlap(name):
    ev = []
    for s, e, q in by[name]:
        ev.append((s, 1)); ev.append((e, -1))
    ev.sort()
    cur = mx = 0
    for _, d in ev:
        cur += d; mx = max(mx, cur)
    return mx
for n, v in sorted(by.items(), key=lambda kv: -sum(e - s for s, e, _ in kv[1]))[:14]:
    d = [(e - s) / 1e3 for s, e, _ in v]
    d_sorted = sorted(d)
    print("%-28s n %5d  median %8.1f us  p90 %8.1f  max %9.1f  in flight at once: %d" % (n, len(d), d_sorted[len(d) // 2], d_sorted[int(len(d) * 0.9)], d_sorted[-1], overlap(n)))
PY
