#!/bin/bash
# Where an ICP step's time goes on the GPU: the kernels of the steady-state iterations of the bench's Fits (C4) with
# their start relative to the iteration's grid kernel, duration and the gap in front -- medians over the iterations.
#   bash tools/icp_timeline.sh [tag]
TAG=${1:-icptl}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 bench.py --steps 200 --warmup 40 --no-cpu-baseline --no-extras > $OUT/bench.log 2>&1
python3 - "$OUT" <<'PY' | tee $OUT/timeline.txt
import csv, glob, sys, statistics
rows = []
for p in glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("<")[0][-28:]))
rows.sort()
its, cur = [], []
for s, e, name in rows:
    if "icp_grid_kernel" in name:
        if cur: its.append(cur)
        cur = []
    cur.append((s, e, name))
# steady-state iterations: the same kernels in the same order as the most common pattern
from collections import Counter
pat = Counter(tuple(n for _, _, n in it) for it in its).most_common(1)[0][0]
its = [it for it in its if tuple(n for _, _, n in it) == pat]
print("iterations", len(its), "kernels per iteration", len(pat))
for i, name in enumerate(pat):
    st = statistics.median((it[i][0] - it[0][0]) / 1e3 for it in its)
    du = statistics.median((it[i][1] - it[i][0]) / 1e3 for it in its)
    gap = statistics.median(((it[i][0] - max(x[1] for x in it[:i])) / 1e3 if i else 0.0) for it in its)
    print("%-30s start %7.1f  dur %7.1f  end %7.1f  gap behind the kernels before %6.1f" % (name, st, du, st + du, gap))
# iteration to iteration (consecutive in the trace)
starts = sorted(it[0][0] for it in its)
d = [(b - a) / 1e3 for a, b in zip(starts, starts[1:]) if (b - a) / 1e3 < 200]
print("grid kernel to grid kernel: median %.1f us" % statistics.median(d))
PY
rm -rf $OUT/trace
