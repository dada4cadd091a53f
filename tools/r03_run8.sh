#!/bin/bash
# round 3, eighth GPU call: two-rows-per-lane kernel as Cholesky entry point and fused pipeline (16 < n <= 25), A/B + the suite
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03h
mkdir -p $O
cd $R
echo "== correctness of the touched paths ==" | tee $O/log.txt
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "cholesky or pipeline or spd" > $O/pytest_sub.txt 2>&1
echo "rc=$?" | tee -a $O/log.txt; tail -12 $O/pytest_sub.txt | tee -a $O/log.txt
for sw in 1 0; do
  echo "-- MATINV_ROWLANE2_SPD=$sw" | tee -a $O/log.txt
  MATINV_ROWLANE2_SPD=$sw timeout -k 10 200 python3 tools/time_sizes.py f64 chol 17 20 24 25 2>&1 | grep "n=" | tee -a $O/log.txt
  MATINV_ROWLANE2_SPD=$sw timeout -k 10 200 python3 tools/time_gp_sizes.py f64 17 20 24 25 2>&1 | grep "n=" | tee -a $O/log.txt
  MATINV_ROWLANE2_SPD=$sw timeout -k 10 200 python3 tools/time_sizes.py f32 chol 20 24 2>&1 | grep "n=" | tee -a $O/log.txt
  MATINV_ROWLANE2_SPD=$sw timeout -k 10 200 python3 tools/time_gp_sizes.py f32 20 24 2>&1 | grep "n=" | tee -a $O/log.txt
done
echo "== full gpu tests ==" | tee -a $O/log.txt
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1
echo "rc=$?" | tee -a $O/log.txt; tail -6 $O/pytest_gpu.txt | tee -a $O/log.txt
