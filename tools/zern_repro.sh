#!/bin/bash
# Dev build for the Zernike wrong-result hunt: the library with the HISTORICAL
# configuration -- every radial order fully unrolled, q-recursion constants as immediates, 256-register
# budget -- into build/zrepro/ (git-ignored; travels with gpurun).  Use with PXF_LIB=build/zrepro/libpxf_hip.so
# and tools/zern_probe.py.  Extra -D flags may be appended:  tools/zern_repro.sh tag -DFOO=1
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TAG=${1:-zrepro}; shift || true
OUT=$ROOT/build/$TAG
mkdir -p "$OUT"
DEFS="-DPXF_ZERN_LOOP_FROM=16 -DPXF_ZERN_HLDS_FROM=99 -DPXF_ZERN_MIN_WAVES_ALL=2 $*"
make -C "$ROOT/pyxfocus_amd/csrc" -j8 OBJDIR="$OUT" LIB="$OUT/libpxf_hip.so" "EXTRA=$DEFS" 2>&1 | tail -2
ls -la "$OUT/libpxf_hip.so"
