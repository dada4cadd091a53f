#!/bin/bash
# Kernel-only throughput of every entry point over the whole size range (device-resident batches of ~1.6 GB or 100 k items,
# SPD inputs, median of 5 launches) -> one text table. usage (inside gpurun): bash tools/size_sweep.sh > gpurun_out/size_sweep.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
S="2 4 8 12 16 20 24 32 40 48 50 56 60 64 72 80 96 100 112 120 128 130 144 160 176 177 192 200 256 384 512 768 1024"
echo "# size sweep: python tools/time_sizes.py <dtype> <algo> n...   (kernel-only, SPD inputs R + R^T + n I)"
for dt in f64 f32; do for algo in gj chol; do echo "## inversion $dt $algo"; python3 $R/tools/time_sizes.py $dt $algo $S 2>/dev/null | grep "n="; done; done
for dt in f64 f32; do
  echo "## inversion $dt gj, GENERAL U(0,1) input (needs row exchanges)"
  MATINV_TIME_GENERAL=1 python3 $R/tools/time_sizes.py $dt gj 8 16 20 32 40 48 50 64 72 80 96 100 128 130 160 192 200 224 256 384 512 768 1024 2>/dev/null | grep "n="
done
for dt in f64 f32; do echo "## fused mean pipeline $dt"; python3 $R/tools/time_gp_sizes.py $dt $S 2>/dev/null | grep "n="; done
