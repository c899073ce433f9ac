#!/bin/bash
# The general path's step on the GPU's own clock for a MID-SIZE cloud (tools/small_vs_general.py's random surface): every
# kernel of three iterations in the middle of the last Fit -- start, duration, the gap in front:  bash tools/general_timeline.sh 16000
N=${1:-16000}
OUT=$PWD/gpurun_out/general_timeline
mkdir -p $OUT; rm -rf $OUT/trace
export TMPDIR=/tmp
cat > $OUT/run.py <<PY
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from pcgol_amd import icp, kdtree, synth
n = $N
c = synth.c4_icp(n=n, width=2.0 + n / 8000.0)
w, th = np.full(6, 0.3, np.float32), np.full(6, -1.0, np.float32)
t = kdtree.New(c["base"])
reg = icp.PointToPointICPGradient(icp.PointToPointEvaluator(icp.NearestPointCorresponder(MaxDist=0.5), MinPairs=6),
                                  icp.GradientDescentUpdaterFactory(Weight=w, Threshold=th, MaxIteration=20))
for _ in range(5):
    reg.Fit(t, c["target"])
PY
PCGX_ICP_SMALL=0 timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $OUT/run.py > $OUT/run.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
ev = []
for p in glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:]))
ev.sort()
last_chain = [i for i, e in enumerate(ev) if "strict_chain" in e[2]]
i1 = last_chain[-8]
i0 = last_chain[-11] + 1
prev = None
for e in ev[i0:i1 + 1]:
    print("%9.2f us  %7.2f us  gap %6.2f  %s" % ((e[0] - ev[i0][0]) / 1e3, (e[1] - e[0]) / 1e3, (e[0] - prev) / 1e3 if prev else 0.0, e[2]))
    prev = e[1]
PY
