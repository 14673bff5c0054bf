#!/bin/bash
# tools/build_variant.sh NAME SOURCE.hip "-DFLAG ..."  ->  build_variants/lib_NAME.so : the in-tree library with ONE translation unit recompiled
# with extra flags (same-box A/B of kernel variants through MFX_LIB_PATH).  Run python -m monoflex_amd.build first.
set -e
cd "$(dirname "$0")/.."
NAME=$1; SRC=$2; EXTRA=$3
mkdir -p build_variants/obj
O=build_variants/obj/${NAME}_${SRC%.hip}.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $EXTRA -c monoflex_amd/csrc/$SRC -o $O
OBJS=$(ls monoflex_amd/csrc/build/*.o | grep -v "/${SRC%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_variants/lib_${NAME}.so $OBJS $O
echo build_variants/lib_${NAME}.so
