#!/bin/bash
# Builds the library of another commit into experiments/ab/libpcgx_head.so (what tools/ab.sh compares the tree's with):
#   bash tools/mk_ab.sh [git-ref]      (default HEAD: the committed state against the working tree's edits)
ref=${1:-HEAD}
tmp=$(mktemp -d /tmp/pcgx_ab.XXXXXX)
git archive "$ref" pcgol_amd include | tar -x -C "$tmp"
(cd "$tmp" && python -c "
import sys; sys.path.insert(0, '.')
from pcgol_amd import build; build.build(force=True)") || exit 1
mkdir -p experiments/ab
cp "$tmp/pcgol_amd/libpcgx.so" experiments/ab/libpcgx_head.so
rm -rf "$tmp"
ls -la experiments/ab/libpcgx_head.so
