"""Condenses a profiles/collect.sh run (rocprofv3 CSVs under gpurun_out/prof_<tag>) into
profiles/<tag>_kernel_stats.csv, profiles/<tag>_pmc.json and profiles/<tag>_fetch_probe.json
(small, committed)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

# FETCH_SIZE on gfx950 (tools/fetch_probe.py, <tag>_fetch_probe.json): a wide coalesced streaming read is
# tallied at HALF its bytes (copy of 1 GiB: 512 MiB reported), a random 4-byte gather at 64 bytes per
# element (the sector it costs; 49 G elements/s = 3.1 TB/s of sectors), a random 16-byte row at 64-128
# bytes.  So: streaming kernels x 2; kernels whose loads are mostly scattered gathers x 1, plus half
# of their (known) coalesced reads, which the counter under-reports like any streaming read.
GATHER_KERNELS = ("icp_grid_kernel", "grid_nearest_kernel", "seg_reduce_kernel", "icp_corr_kernel",
                  "nearest_kernel", "range_kernel")
# coalesced bytes the gather kernels read per launch at the bench's sizes (1M targets / queries, 10M voxel points)
STREAMED = {"icp_grid_kernel<false": (12 + 16 * 19 / 20) * 1e6,          # target xyz + previous pair (19 of 20 iterations)
            "icp_corr_kernel<false, false, true, false>": 28e6,           # strict sessions: the tile sums stream the caller-order pairs + targets
            "icp_grid_kernel<true": (12 + 16 * 19 / 20 + 4) * 1e6,       # + matched id
            "grid_nearest_kernel": (12 + 4) * 1e6,                        # query + its position in the batch
            "seg_reduce_kernel": 8 * 10e6}                                # sorted key + sorted index
# one VoxelGrid C3 call (plain mode, 22-bit key: three radix passes) in kernel launches
VOXEL_CALL = {"minmax_partial_packed_kernel": 1, "minmax_final_kernel": 1, "voxel_key_kernel": 1,
              "rs_hist_kernel<16>": 3, "rs_scan_rows_kernel": 3, "rs_scatter_kernel<16>": 3,
              "seg_count_kernel": 1, "seg_scan_kernel": 1, "seg_reduce_kernel": 1}


def short(name):
    n = name.split("(")[0]
    for p in ("void ", "pcgx::"):
        n = n.replace(p, "")
    return n.strip()


# kernels that bench.py launches first on one kind of input, then as often on another: counters per half
HALVES = {"grid_nearest_kernel<false>": ("queries in Morton order", "queries in caller order")}


def read_pmc(pattern):
    pmc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for p in glob.glob(pattern, recursive=True):
        rows = list(csv.DictReader(open(p)))
        order = defaultdict(lambda: defaultdict(list))  # kernel -> counter -> values in launch order
        for r in rows:
            k = short(r.get("Kernel_Name", "?"))
            v = float(r.get("Counter_Value", 0) or 0)
            a = pmc[k][r.get("Counter_Name")]
            a[0] += v
            a[1] += 1
            if k in HALVES:
                order[k][r.get("Counter_Name")].append((int(r.get("Dispatch_Id", 0) or 0), v))
        for k, cs in order.items():
            for c, vals in cs.items():
                vals.sort()
                h = len(vals) // 2
                for part, name in ((vals[:h], HALVES[k][0]), (vals[h:], HALVES[k][1])):
                    a = pmc["%s [%s]" % (k, name)][c]
                    a[0] += sum(v for _, v in part)
                    a[1] += len(part)
    return pmc


def main():
    out, tag = sys.argv[1], sys.argv[2]
    here = os.path.dirname(os.path.abspath(__file__))
    stats = glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        rows = list(csv.reader(open(stats[0])))
        with open(os.path.join(here, "%s_kernel_stats.csv" % tag), "w", newline="") as f:
            csv.writer(f).writerows(rows)
    # ---- the byte counters on known patterns: factor = true bytes / (counter x 1024)
    probe = {}
    pp = read_pmc(os.path.join(out, "probe_*", "**", "*counter_collection.csv"))
    known = {"copy": (1 << 30, 1 << 30), "gather16": ((1 << 24) * 16, (1 << 24) * 16), "gather4": ((1 << 26) * 4, (1 << 26) * 4)}
    for k, cs in pp.items():
        low = k.lower()
        which = None
        if "index" in low or "gather" in low:
            # two index_select launches per repetition: the 16-byte rows one writes 256 MiB, the 4-byte one too;
            # told apart by the fetch volume (rows touch 16 B of a 64-B sector, elements 4 B)
            which = "gather"
        elif "copy" in low or "memcpy" in low.replace("_", ""):
            which = "copy"
        if which:
            probe.setdefault(which, {})[k] = {c: a[0] / max(a[1], 1) for c, a in cs.items()}
    fetch_scale_stream, fetch_scale_gather = 2.0, 1.0
    rep = {"kernels": probe}
    for k, cs in probe.get("copy", {}).items():
        if cs.get("FETCH_SIZE", 0) > 0 and cs["FETCH_SIZE"] * 1024 > (1 << 28):
            fetch_scale_stream = (1 << 30) / (cs["FETCH_SIZE"] * 1024)
            rep["copy"] = {"kernel": k, "true_read_bytes": 1 << 30, "FETCH_SIZE_KiB": cs["FETCH_SIZE"],
                           "bytes_per_reported_byte": fetch_scale_stream,
                           "WRITE_SIZE_KiB": cs.get("WRITE_SIZE"), "true_write_bytes": 1 << 30}
    g = sorted(probe.get("gather", {}).items(), key=lambda kv: -kv[1].get("FETCH_SIZE", 0))
    if g:
        # random rows: every row costs at least one 64-byte sector; report the counter per row / element
        rep["gather"] = [{"kernel": k, "FETCH_SIZE_KiB": cs.get("FETCH_SIZE"), "WRITE_SIZE_KiB": cs.get("WRITE_SIZE"),
                          "reported_bytes_per_gathered_item": cs.get("FETCH_SIZE", 0) * 1024.0 /
                          ((1 << 24) if "vectorized_gather" in k else (1 << 26)),
                          "item_bytes": 16 if "vectorized_gather" in k else 4} for k, cs in g]
    rep["fetch_scale_streaming"] = fetch_scale_stream
    rep["fetch_scale_gather"] = fetch_scale_gather
    with open(os.path.join(here, "%s_fetch_probe.json" % tag), "w") as f:
        json.dump(rep, f, indent=1, sort_keys=True)
    # ---- PMC of the bench: average per dispatch per kernel
    pmc = read_pmc(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"))
    summary = {k: {c: {"mean_per_dispatch": a[0] / max(a[1], 1), "dispatches": a[1]} for c, a in cs.items()}
               for k, cs in pmc.items()}
    for k in list(summary):
        gather = any(gk in k for gk in GATHER_KERNELS)
        summary[k]["fetch_scale"] = 1.0 if gather else fetch_scale_stream
        summary[k]["fetch_add_bytes"] = 0.0
        for name, b in STREAMED.items():
            if gather and k.startswith(name):
                summary[k]["fetch_add_bytes"] = b / 2
    vox = {"FETCH_SIZE": {"mean_per_dispatch": 0.0, "dispatches": 1}, "WRITE_SIZE": {"mean_per_dispatch": 0.0, "dispatches": 1},
           "fetch_scale": 1.0, "fetch_add_bytes": 0.0, "kernels": VOXEL_CALL}
    ok = True
    for k, times in VOXEL_CALL.items():
        if k not in summary or "FETCH_SIZE" not in summary[k] or "WRITE_SIZE" not in summary[k]:
            ok = False
            continue
        vox["FETCH_SIZE"]["mean_per_dispatch"] += times * (summary[k]["FETCH_SIZE"]["mean_per_dispatch"] * summary[k]["fetch_scale"] +
                                                            summary[k]["fetch_add_bytes"] / 1024.0)
        vox["WRITE_SIZE"]["mean_per_dispatch"] += times * summary[k]["WRITE_SIZE"]["mean_per_dispatch"]
    if ok:
        summary["voxel_pipeline (one C3 call, sum over its kernels, fetch already scaled)"] = vox
    # the build the counters were collected on (bench.py compares it with the build it times)
    sys.path.insert(0, os.path.dirname(here))
    from pcgol_amd import build
    summary["_build"] = {"source_hash": build.source_hash(), "tag": tag}
    with open(os.path.join(here, "%s_pmc.json" % tag), "w") as f:
        json.dump(summary, f, indent=1, sort_keys=True)
    print("wrote", tag, "kernels with PMC:", len(summary), "fetch scale streaming %.3f" % fetch_scale_stream)


if __name__ == "__main__":
    main()
