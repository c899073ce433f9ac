"""Reads a PCGX_STRICT_TRACE dump of tools/strict_trace_run.py (no clock reads inside the walk): how long the chain kernel's
rows are in the kernel, over the iterations of the Fit."""
import sys
import numpy as np
blocks = open(sys.argv[1]).read().split("#\n")
per = []
for bi in range(len(blocks) - 1):
    a = np.array([list(map(int, l.split())) for l in blocks[bi].strip().split("\n")], dtype=np.uint64)
    per.append([(int(a[r, 15]) - int(a[r, 11])) / 100 for r in range(8)])
per = np.array(per)
parts = []
for bi in range(len(blocks) - 1):
    a = np.array([list(map(int, l.split())) for l in blocks[bi].strip().split("\n")], dtype=np.uint64)
    parts.append([[(int(a[r, 12]) - int(a[r, 11])) / 100, (int(a[r, 13]) - int(a[r, 12])) / 100, (int(a[r, 15]) - int(a[r, 13])) / 100] for r in range(8)])
parts = np.array(parts)
print("  mean per row: records arrive %.1f us after the row's workgroup came, runs composed %.1f us later, walk %.1f us; the slowest row's walk: median %.1f" % (
    parts[:, :, 0].mean(), parts[:, :, 1].mean(), parts[:, :, 2].mean(), np.median(parts[:, :, 2].max(axis=1))))
print("rows in the kernel, us: mean over iterations and rows %.1f; slowest row of an iteration: mean %.1f, median %.1f, max %.1f" % (
    per.mean(), per.max(axis=1).mean(), np.median(per.max(axis=1)), per.max()))
print("  per iteration (slowest row):", " ".join("%.0f" % v for v in per.max(axis=1)))
