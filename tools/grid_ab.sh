#!/bin/bash
# A/B timing of the grid kernels: kNN C2 call (presorted / caller order) and the ICP step (f64-tree, strict)
PROBE_ONLY= timeout -k 10 200 python tools/grid_probe.py 2>&1 | grep -E "grid presort|grid unsorted|agree"
timeout -k 10 200 python tools/strict_probe.py 2>&1 | grep -E "^strict [01]|MISMATCH" | tail -3
