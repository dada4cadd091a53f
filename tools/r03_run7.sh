#!/bin/bash
# round 3, seventh GPU call: one-wavefront pivoting kernel for 64 < n <= 96 (A/B), then the full suite and the default bench line
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03g
mkdir -p $O
cd $R
echo "== correctness of the pivoting paths ==" | tee $O/log.txt
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "general or pivot or tile_family or singular or square or default_policy" > $O/pytest_sub.txt 2>&1
echo "rc=$?" | tee -a $O/log.txt; tail -4 $O/pytest_sub.txt | tee -a $O/log.txt
for sw in 1 0; do
  echo "-- MATINV_TILEP_ONEWAVE=$sw: f64 general, pivoting kernel forced" | tee -a $O/log.txt
  MATINV_TILEP_ONEWAVE=$sw MATINV_TIME_GENERAL=1 MATINV_TIME_KERNEL=tilep timeout -k 10 200 python3 tools/time_sizes.py f64 gj 72 80 88 96 2>&1 | grep "n=" | tee -a $O/log.txt
done
echo "== full gpu tests ==" | tee -a $O/log.txt
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1
echo "rc=$?" | tee -a $O/log.txt; tail -6 $O/pytest_gpu.txt | tee -a $O/log.txt
