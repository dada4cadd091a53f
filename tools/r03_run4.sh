#!/bin/bash
# round 3, fourth GPU call: binary-tree pivot-row gather (tilep4 / tilepw / tilepb) A/B, two-level blocked GJ below n = 384
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03d
mkdir -p $O
cd $R
echo "== correctness, default kernels (tree gather) ==" | tee $O/log.txt
timeout -k 10 500 python3 -m pytest tests/test_gpu_parity.py -x -q -k "general or pivot or tilep or singular or square or default_policy" > $O/pytest_tree.txt 2>&1
echo "rc=$?" | tee -a $O/log.txt; tail -4 $O/pytest_tree.txt | tee -a $O/log.txt
echo "== correctness, blk variant ==" | tee -a $O/log.txt
MATINV_TILEP_WAVES=blk timeout -k 10 500 python3 -m pytest tests/test_gpu_parity.py -x -q -k "general or pivot or tilep or singular or square" > $O/pytest_blk.txt 2>&1
echo "rc=$?" | tee -a $O/log.txt; tail -4 $O/pytest_blk.txt | tee -a $O/log.txt
echo "== A/B general (pivoting kernel forced) ==" | tee -a $O/log.txt
for w in 4 blk; do
  echo "-- f64 MATINV_TILEP_WAVES=$w" | tee -a $O/log.txt
  MATINV_TILEP_WAVES=$w MATINV_TIME_GENERAL=1 MATINV_TIME_KERNEL=tilep timeout -k 10 300 python3 tools/time_sizes.py f64 gj 32 64 72 80 96 100 112 128 130 144 160 192 2>&1 | grep "n=" | tee -a $O/log.txt
done
for w in 4 blk; do
  echo "-- f32 MATINV_TILEP_WAVES=$w" | tee -a $O/log.txt
  MATINV_TILEP_WAVES=$w MATINV_TIME_GENERAL=1 MATINV_TIME_KERNEL=tilep timeout -k 10 300 python3 tools/time_sizes.py f32 gj 80 96 128 160 256 2>&1 | grep "n=" | tee -a $O/log.txt
done
echo "== blocked GJ: two-level scheme from n = 200 instead of 384 ==" | tee -a $O/log.txt
for m in 384 200; do
  echo "-- MATINV_BGJ_TWO_LEVEL_MIN=$m" | tee -a $O/log.txt
  MATINV_BGJ_TWO_LEVEL_MIN=$m MATINV_TIME_GENERAL=1 timeout -k 10 300 python3 tools/time_sizes.py f64 gj 200 256 320 384 2>&1 | grep "n=" | tee -a $O/log.txt
  MATINV_BGJ_TWO_LEVEL_MIN=$m MATINV_TIME_GENERAL=1 timeout -k 10 300 python3 tools/time_sizes.py f32 gj 320 2>&1 | grep "n=" | tee -a $O/log.txt
done
