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
GATHER_KERNELS = ("icp_grid_kernel", "grid_nearest_kernel", "grid_nearest_rec_kernel", "seg_reduce_kernel", "icp_corr_kernel",
                  "nearest_kernel", "range_kernel")
# coalesced bytes the gather kernels read per launch at the bench's sizes (1M targets / queries, 10M voxel points)
STREAMED = {"icp_grid_kernel<false": (12 + 16 * 19 / 20) * 1e6,          # target xyz + previous pair (19 of 20 iterations)
            "icp_grid_kernel<true": (12 + 16 * 19 / 20 + 4) * 1e6,       # + matched id
            "grid_nearest_kernel": 12 * 1e6,                              # queries in the caller's order
            "grid_nearest_rec_kernel": 16 * 1e6,                          # {x, y, z, index} records, cell after cell
            "seg_reduce_kernel": 8 * 10e6}                                # sorted key + sorted index
# One VoxelGrid C3 call in kernel launches: the bucket path (csrc/voxel_bucket.hip).  bench.py runs the plain and the
# chunked filter as often as each other; every kernel below is launched once per call of either (scan_rows twice),
# and by nothing else in the bench, so a call's bytes = the kernels' totals over all their dispatches / the number of
# calls (= dispatches of the bucket kernel).  A kernel that is listed here and absent from the counters means the
# pipeline has changed under this list: the summary then FAILS instead of leaving the row out (round 3's BENCH line
# carried `"traffic": null` for the filter because of exactly that).
VOXEL_KERNELS = ("minmax_partial_packed_kernel", "vb_key_hist_kernel", "vb_scan_rows_kernel", "vb_scatter_kernel<true, false>",
                 "vb_hist2_kernel", "vb_scatter_kernel<false, false>", "vb_bucket_kernel<false>", "vb_place_kernel<false>")
VOXEL_CALLS_BY = "vb_bucket_kernel<false>"


def short(name):
    n = name.split("(")[0]
    for p in ("void ", "pcgx::"):
        n = n.replace(p, "")
    return n.strip()


# kernels that bench.py launches first on one kind of input, then as often on another: counters per half
HALVES = {}   # (round 3: grid_nearest_kernel ran on Morton-ordered, then on caller-ordered queries; the ordered batches have a kernel of their own now)


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
    if any(k.startswith("vb_") or k.startswith("seg_reduce") for k in summary):  # (a run with the extras)
        missing = [k for k in VOXEL_KERNELS if k not in summary or "FETCH_SIZE" not in summary[k] or "WRITE_SIZE" not in summary[k]]
        if missing:
            raise SystemExit("profiles/summarize.py: the VoxelGrid call no longer launches %s -- VOXEL_KERNELS is stale, "
                             "bench.py's voxel traffic row would silently disappear" % ", ".join(missing))
        calls = summary[VOXEL_CALLS_BY]["FETCH_SIZE"]["dispatches"]
        vox = {"FETCH_SIZE": {"mean_per_dispatch": 0.0, "dispatches": calls}, "WRITE_SIZE": {"mean_per_dispatch": 0.0, "dispatches": calls},
               "fetch_scale": 1.0, "fetch_add_bytes": 0.0, "kernels": {}}
        for k in VOXEL_KERNELS:
            f, w = summary[k]["FETCH_SIZE"], summary[k]["WRITE_SIZE"]
            fetch = (f["mean_per_dispatch"] * summary[k]["fetch_scale"] + summary[k]["fetch_add_bytes"] / 1024.0) * f["dispatches"] / calls
            write = w["mean_per_dispatch"] * w["dispatches"] / calls
            vox["FETCH_SIZE"]["mean_per_dispatch"] += fetch
            vox["WRITE_SIZE"]["mean_per_dispatch"] += write
            vox["kernels"][k] = {"launches_per_call": f["dispatches"] / calls, "fetch_KiB_per_call": fetch, "write_KiB_per_call": write}
        summary["voxel_pipeline (one C3 call -- plain and chunked averaged --, sum over its kernels, fetch already scaled)"] = vox
    # the build the counters were collected on (bench.py compares it with the build it times)
    sys.path.insert(0, os.path.dirname(here))
    from pcgol_amd import build
    summary["_build"] = {"source_hash": build.source_hash(), "tag": tag}
    with open(os.path.join(here, "%s_pmc.json" % tag), "w") as f:
        json.dump(summary, f, indent=1, sort_keys=True)
    print("wrote", tag, "kernels with PMC:", len(summary), "fetch scale streaming %.3f" % fetch_scale_stream)


if __name__ == "__main__":
    main()
