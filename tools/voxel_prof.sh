#!/bin/bash
# Per-kernel timing of the voxel filter (C3 shape) on the GPU box:  bash tools/voxel_prof.sh [tag]
TAG=${1:-vox}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/voxel_probe.py > $OUT/probe.log 2>&1
tail -2 $OUT/probe.log
python3 - "$OUT" <<'PY'
import csv, glob, sys
for p in glob.glob(sys.argv[1] + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        print("%-28s calls %4s avg %9.1f us  total %9.1f us" % (r["Name"].split("(")[0][-28:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
PY
