#!/bin/bash
# One host-pointer Fit of the reference's benchmark shape on the GPU's own clock: every kernel and copy of the LAST Fit of
# tools/small_fit_probe.py, start (us from the first), duration, gap in front:  bash tools/fit_timeline.sh 1024 [tag]
N=${1:-1024}; TAG=${2:-fit_timeline}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT; rm -rf $OUT/trace
export TMPDIR=/tmp
PCGX_PROBE_HOST_ONLY=1 timeout -k 10 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/trace -- python3 tools/small_fit_probe.py $N 6 > $OUT/run.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
ev = []
for p in glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-46:]))
for p in glob.glob(sys.argv[1] + "/trace/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "") + " " + r.get("Size", "")))
ev.sort()
# the last Fit: from the last upload in front of the last icp_small_fit_kernel
last = max(i for i, e in enumerate(ev) if "icp_small_fit" in e[2])
first = last
while first > 0 and ev[first][0] - ev[first - 1][1] < 60000 and "icp_small_fit" not in ev[first - 1][2]:
    first -= 1
t0 = ev[first][0]
prev = None
for e in ev[first:last + 3]:
    print("%9.2f us  %8.2f us  gap %7.2f  %s" % ((e[0] - t0) / 1e3, (e[1] - e[0]) / 1e3, (e[0] - prev) / 1e3 if prev else 0.0, e[2]))
    prev = e[1]
PY
