#!/bin/bash
# PMC counters of any python tool (GPU box):  bash tools/pmc_any.sh <tag> "<counters>" tools/x.py [args...]
TAG=$1; PMC=$2; shift; shift
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 90 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/pmc -- python3 "$@" > $OUT/run.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for p in glob.glob(sys.argv[1] + "/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        a = acc[r["Kernel_Name"].split("(")[0][-40:]][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
for k, cs in acc.items():
    print(k, {c: round(a[0] / a[1]) for c, a in cs.items()}, "dispatches", max(a[1] for a in cs.values()))
PY
