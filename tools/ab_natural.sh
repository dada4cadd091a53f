#!/bin/bash
# A/B of the natural-order 64x64 kernels on one box: r01 kernel (MATINV_TILE_NATURAL=old) vs the lane-per-row panel kernel
for i in 1 2; do
  MATINV_GJ_POLICY=natural python3 tools/time_kernel.py 64 100000 0 21 | sed 's/^/old: /'
  MATINV_GJ_POLICY=natural MATINV_TILE_NATURAL=new python3 tools/time_kernel.py 64 100000 0 21 | sed 's/^/new: /'
done
MATINV_GJ_POLICY=natural python3 tools/time_kernel.py 32 400000 0 21 | sed 's/^/old: /'
MATINV_GJ_POLICY=natural MATINV_TILE_NATURAL=new python3 tools/time_kernel.py 32 400000 0 21 | sed 's/^/new: /'
MATINV_GJ_POLICY=natural python3 tools/time_kernel.py 48 170000 0 21 | sed 's/^/old: /'
MATINV_GJ_POLICY=natural MATINV_TILE_NATURAL=new python3 tools/time_kernel.py 48 170000 0 21 | sed 's/^/new: /'
MATINV_GJ_POLICY=natural python3 tools/time_kernel.py 64 200000 0 21 f32 | sed 's/^/old: /'
MATINV_GJ_POLICY=natural MATINV_TILE_NATURAL=new python3 tools/time_kernel.py 64 200000 0 21 f32 | sed 's/^/new: /'
