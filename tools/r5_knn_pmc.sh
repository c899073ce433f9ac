#!/bin/bash
# PMC counters of the kNN C2 search kernel, one query per lane against four lanes per query (PCGX_KNN_COOP=1):
# separate --pmc passes (kernel-trace only), per-kernel means of grid_nearest_rec_kernel
tag=${1:-r5knnpmc}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for mode in 0 1; do
  i=0
  for PASS in "TCP_TOTAL_ACCESSES_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TD_TD_BUSY_sum TA_TA_BUSY_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" "GRBM_GUI_ACTIVE SQ_WAVES" "FETCH_SIZE" ; do
    i=$((i+1))
    PCGX_KNN_COOP=$mode timeout -k 10 120 rocprofv3 --kernel-trace --pmc $PASS --output-format csv -d gpurun_out/${tag}_m${mode}_p$i -- python3 tools/knn_time.py > gpurun_out/${tag}_m${mode}_p$i.log 2>&1
    echo "mode $mode pass $i rc=$?"
  done
done
python3 - "$tag" <<'PY'
import csv, glob, json, sys
from collections import defaultdict
tag = sys.argv[1]
out = {}
for mode in (0, 1):
    acc = defaultdict(lambda: [0.0, 0])
    for p in glob.glob("gpurun_out/%s_m%d_p*/**/*counter_collection.csv" % (tag, mode), recursive=True):
        for r in csv.DictReader(open(p)):
            if "grid_nearest_rec_kernel" not in r["Kernel_Name"]:
                continue
            a = acc[r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
    out["four lanes per query (PCGX_KNN_COOP=1)" if mode else "one query per lane (default)"] = {k: v[0] / v[1] for k, v in sorted(acc.items())}
json.dump({"_about": "per-launch means of grid_nearest_rec_kernel on kNN C2 (1M queries x 1M-point tree), rocprofv3 --pmc, tools/r5_knn_pmc.sh",
           "counters": out}, open("gpurun_out/%s.json" % tag, "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf gpurun_out/${tag}_m*_p*/
