#!/bin/bash
# Host-side AddressSanitizer pass over the C ABI's HOST code (program planner, parameter blocks,
# Zernike / Legendre table folding) -- GPU ASan is not available on this pool, and none of this
# needs a GPU: launches simply fail with PXF_EHIP after the host work is done.
# usage: bash tools/asan_host.sh        (builds into build/asan, runs tools/asan_host.py, removes the build)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
make -C "$ROOT/pyxfocus_amd/csrc" -j8 OBJDIR="$ROOT/build/asan" LIB="$ROOT/build/asan/libpxf_hip_asan.so" \
     EXTRA="-Xarch_host -fsanitize=address -Xarch_host -fno-omit-frame-pointer -g" > /dev/null
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
PXF_LIB="$ROOT/build/asan/libpxf_hip_asan.so" LD_PRELOAD="$RT" ASAN_OPTIONS=detect_leaks=0 python3 "$ROOT/tools/asan_host.py"
rm -rf "$ROOT/build/asan"
