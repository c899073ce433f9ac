#!/bin/bash
# The ICP grid pass of the last Fit of a short bench run, iteration by iteration, for the tree's library and for
# variants (tools/mk_variant.sh):  bash tools/icp_grid_times.sh tag [variant ...]
tag=$1; shift
for v in tree "$@"; do
  lib=""; [ "$v" != tree ] && lib="PCGX_LIB=experiments/ab/libpcgx_$v.so"
  env $lib bash tools/prof_any.sh ${tag}_$v bench.py --steps 400 --warmup 100 --no-cpu-baseline --no-extras > /dev/null 2>&1
  python3 - "$tag" "$v" <<'PY'
import csv, glob, os, sys
ps = sorted(glob.glob("gpurun_out/%s_%s/trace/**/*kernel_trace.csv" % (sys.argv[1], sys.argv[2]), recursive=True), key=os.path.getmtime)
allr = [r for r in csv.DictReader(open(ps[-1]))]
for name in ("icp_grid_kernel<false, false, false>", "icp_grid4_kernel", "icp_corr_kernel<false, false, true, false"):
    rows = sorted((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in allr if name in r["Kernel_Name"])
    print("%-8s %-40s" % (sys.argv[2], name[:40]), " ".join("%.1f" % v[1] for v in rows[-20:]))
PY
done
