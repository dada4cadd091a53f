#!/bin/bash
# round 3, call 15: blocked paths after the occupancy retune -- parity of everything blocked, rates, kernel split at 256 / 1024
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03o
mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "large_n or blocked or chunk or first_pass" > $O/pytest.txt 2>&1
echo "rc=$?" | tee $O/log.txt; tail -3 $O/pytest.txt | tee -a $O/log.txt
MATINV_TIME_GENERAL=1 timeout -k 10 200 python3 tools/time_sizes.py f64 gj 193 200 224 256 320 384 512 768 1024 2>&1 | grep "n=" | tee -a $O/log.txt
MATINV_TIME_GENERAL=1 timeout -k 10 200 python3 tools/time_sizes.py f32 gj 257 300 384 512 1024 2>&1 | grep "n=" | tee -a $O/log.txt
timeout -k 10 200 python3 tools/time_sizes.py f64 chol 200 256 512 1024 2>&1 | grep "n=" | tee -a $O/log.txt
cd /tmp; export TMPDIR=/tmp
for w in gj256g gj1024g; do
  T=/tmp/kt_$w; rm -rf $T; mkdir -p $T
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $T -o a -- python3 $R/bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-others > $O/$w.json 2> $O/$w.err
  python3 $R/tools/rocprof_summary.py $(ls $T/*_results.db | head -1) "$w" > $O/$w.trace.txt 2>&1
  head -11 $O/$w.trace.txt | cut -c1-150 | tee -a $O/log.txt
done
