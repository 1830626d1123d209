#!/bin/bash
# Developer build of the WORKING TREE: build/liballophant_amx_dev.so with -DAMX_DEVELOPER, the only build in which the AMX_*
# A/B switches (AMX_NO_SPLITK, AMX_ATTN_WAVES, AMX_LN_TAP_MAJOR, ...) read the environment.  The product library
# (allophant_amd/liballophant_amx.so) has no such switches.  Use:
#   tools/dev_build.sh && AMX_LIB_PATH=$PWD/build/liballophant_amx_dev.so AMX_NO_SPLITK=1 python tools/geometry_sweep.py f16x3 4:10
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
make -C "$ROOT/allophant_amd/csrc" -j4 DEVELOPER=1 OBJDIR="$ROOT/build/obj_dev" OUT="$ROOT/build/liballophant_amx_dev.so" > /dev/null
ls -la "$ROOT/build/liballophant_amx_dev.so"
