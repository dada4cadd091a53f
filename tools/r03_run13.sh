#!/bin/bash
# round 3, call 13: per-kernel split of the blocked general path at 200 / 256 (and the SPD one at 256), kernel trace only
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03m
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for w in gj256g chol256 gj192g; do
  T=/tmp/kt_$w; rm -rf $T; mkdir -p $T
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $T -o a -- python3 $R/bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-others > $O/$w.json 2> $O/$w.err
  python3 $R/tools/rocprof_summary.py $(ls $T/*_results.db | head -1) "$w" > $O/$w.trace.txt 2>&1
  cat $O/$w.trace.txt | cut -c1-170
done
cd $R
MATINV_TIME_GENERAL=1 timeout -k 10 200 python3 tools/time_sizes.py f64 gj 193 200 208 224 240 256 2>&1 | grep "n=" | tee $O/sizes.txt
