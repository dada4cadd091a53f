#!/bin/bash
# round 3, ninth GPU call: two-wavefront lower-tile SPD sweep (112 < n <= 128, fp64) A/B; one-wave pivoting kernel with the tree gather
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03i
mkdir -p $O
cd $R
echo "== correctness of the touched paths ==" | tee $O/log.txt
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "cholesky or pipeline or spd or general or pivot or square or singular" > $O/pytest_sub.txt 2>&1
echo "rc=$?" | tee -a $O/log.txt; tail -12 $O/pytest_sub.txt | tee -a $O/log.txt
for sw in 1 0; do
  echo "-- MATINV_SPD_TILE2=$sw" | tee -a $O/log.txt
  MATINV_SPD_TILE2=$sw timeout -k 10 200 python3 tools/time_sizes.py f64 chol 113 120 128 2>&1 | grep "n=" | tee -a $O/log.txt
  MATINV_SPD_TILE2=$sw timeout -k 10 200 python3 tools/time_gp_sizes.py f64 113 120 128 2>&1 | grep "n=" | tee -a $O/log.txt
done
echo "-- general n <= 64 with the tree gather (r02 form: 1.60e8 / 4.35e7 at 32 / 64 on this tool's batch sizes)" | tee -a $O/log.txt
MATINV_TIME_GENERAL=1 MATINV_TIME_KERNEL=tilep timeout -k 10 200 python3 tools/time_sizes.py f64 gj 20 32 48 64 2>&1 | grep "n=" | tee -a $O/log.txt
