"""Condenses a profiles/collect.sh run (rocprofv3 CSVs under gpurun_out/prof_<tag>) into
profiles/<tag>_kernel_stats.csv and profiles/<tag>_pmc.json (small, committed)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    out, tag = sys.argv[1], sys.argv[2]
    here = os.path.dirname(os.path.abspath(__file__))
    # 1) kernel stats
    stats = glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        rows = list(csv.reader(open(stats[0])))
        with open(os.path.join(here, "%s_kernel_stats.csv" % tag), "w", newline="") as f:
            csv.writer(f).writerows(rows)
    # 2) PMC: average per dispatch per kernel
    pmc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for p in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(p)):
            k = r.get("Kernel_Name", "?").split("(")[0]
            c = r.get("Counter_Name")
            v = float(r.get("Counter_Value", 0) or 0)
            a = pmc[k][c]
            a[0] += v
            a[1] += 1
    summary = {k: {c: {"mean_per_dispatch": a[0] / max(a[1], 1), "dispatches": a[1]} for c, a in cs.items()}
               for k, cs in pmc.items()}
    with open(os.path.join(here, "%s_pmc.json" % tag), "w") as f:
        json.dump(summary, f, indent=1, sort_keys=True)
    print("wrote", tag, "kernels with PMC:", len(summary))


if __name__ == "__main__":
    main()
