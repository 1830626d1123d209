#!/bin/bash
# Developer A/B: builds the library of a git revision into build/ab/<name>.so (the working tree's library stays where it is),
# so that one gpurun call can time both on the same box (boxes differ by several per cent):
#   tools/ab_build.sh HEAD A && gpurun -- 'for L in build/ab/A.so allophant_amd/liballophant_amx.so; do AMX_LIB_PATH=$PWD/$L python tools/geometry_sweep.py f16x3 32:10; done'
set -e
REV=${1:-HEAD}; NAME=${2:-A}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
git -C "$ROOT" archive "$REV" allophant_amd/csrc include | tar -x -C "$TMP"
mkdir -p "$ROOT/build/ab"
# DEVELOPER=1: the AMX_* A/B switches read the environment only in this build (the product library has none)
make -C "$TMP/allophant_amd/csrc" -j4 DEVELOPER=1 OBJDIR="$TMP/obj" OUT="$ROOT/build/ab/$NAME.so" > /dev/null
rm -rf "$TMP"
ls -la "$ROOT/build/ab/$NAME.so"
