#!/bin/bash
# Profiling only: libmatinv_hip_ldst.so = the library with the natural-order tile kernels (tile_kernels.inc) reduced to their
# loads and stores (-DMATINV_TILE_LDST_ONLY: same access pattern, same launch shape, no elimination). Used through MATINV_LIB to
# price the memory side of the headline kernel by itself (tools/profile_round.sh). Run here (hipcc cross-compiles); the .so travels.
set -e
cd "$(dirname "$0")/../cuda-matrix-inversion_amd"
mkdir -p build_dbg
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-but-set-variable -mllvm -pragma-unroll-threshold=1000000 \
    -DMATINV_TILE_LDST_ONLY -c csrc/tile_kernels.hip -o build_dbg/tile_kernels.o
OBJS=$(ls build/*.o | grep -v "build/tile_kernels.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libmatinv_hip_ldst.so $OBJS build_dbg/tile_kernels.o -lpthread -ldl
ls -la libmatinv_hip_ldst.so
