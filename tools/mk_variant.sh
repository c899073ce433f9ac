#!/bin/bash
# Builds the working tree's library with extra compiler switches into experiments/ab/libpcgx_<name>.so (for PCGX_LIB):
#   bash tools/mk_variant.sh w6 "-DPCGX_KNN_WAVES=6"
name=$1; flags=$2
tmp=$(mktemp -d /tmp/pcgx_var.XXXXXX)
cp -r pcgol_amd include "$tmp"/ && rm -rf "$tmp/pcgol_amd/build" "$tmp/pcgol_amd/libpcgx.so"
(cd "$tmp" && PCGX_EXTRA_CFLAGS="$flags" python -c "
import sys; sys.path.insert(0, '.')
from pcgol_amd import build; build.build(force=True)") || exit 1
mkdir -p experiments/ab
cp "$tmp/pcgol_amd/libpcgx.so" experiments/ab/libpcgx_$name.so
rm -rf "$tmp"
ls -la experiments/ab/libpcgx_$name.so
