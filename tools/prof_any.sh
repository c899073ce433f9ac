#!/bin/bash
# rocprofv3 kernel stats of any python tool:  bash tools/prof_any.sh <tag> tools/x.py [args...]
TAG=$1; shift
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 "$@" > $OUT/run.log 2>&1
tail -2 $OUT/run.log
python3 - "$OUT" <<'PY'
import csv, glob, sys
for p in glob.glob(sys.argv[1] + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        print("%-30s calls %4s avg %9.1f us  total %9.1f us" % (r["Name"].split("(")[0][-30:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
PY
