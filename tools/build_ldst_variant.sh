#!/bin/bash
# Profiling only: libmatinv_hip_ldst.so = the library with the natural-order tile kernels (tile_kernels.inc, the fp64 translation unit
# tile_gj_kernels.hip) reduced to their loads and stores (-DMATINV_TILE_LDST_ONLY: same access pattern, same launch shape, no
# elimination). Used through MATINV_LIB to price the memory side of the headline kernel by itself (tools/profile_round.sh,
# tools/clock_power.sh). Run here (hipcc cross-compiles); the .so travels.
set -e
cd "$(dirname "$0")/../cuda-matrix-inversion_amd"
mkdir -p build_dbg
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-but-set-variable -mllvm -pragma-unroll-threshold=1000000 \
    -DMATINV_TILE_LDST_ONLY -c csrc/tile_gj_kernels.hip -o build_dbg/tile_gj_kernels.o
OBJS=$(for f in csrc/*.hip; do b=build/$(basename $f .hip).o; [ "$b" = build/tile_gj_kernels.o ] || echo $b; done)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libmatinv_hip_ldst.so $OBJS build_dbg/tile_gj_kernels.o -lpthread -ldl
ls -la libmatinv_hip_ldst.so
