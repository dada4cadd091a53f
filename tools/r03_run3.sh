#!/bin/bash
# round 3, third GPU call: the one-barrier-per-tile-column pivoting kernel (tilepb) A/B, the scratch-allocator fix under the
# multi-shard test, then the full suite
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03c
mkdir -p $O
cd $R
echo "== tilepb correctness (general, singular, fixtures) ==" | tee $O/log.txt
MATINV_TILEP_WAVES=blk timeout -k 10 500 python3 -m pytest tests/test_gpu_parity.py -x -q -k "general or pivot or tilep or singular or square or default_policy" > $O/pytest_blk.txt 2>&1
echo "rc=$?" | tee -a $O/log.txt; tail -12 $O/pytest_blk.txt | tee -a $O/log.txt
echo "== A/B general (pivoting kernel forced) ==" | tee -a $O/log.txt
for w in 4 blk; do
  echo "-- f64 MATINV_TILEP_WAVES=$w" | tee -a $O/log.txt
  MATINV_TILEP_WAVES=$w MATINV_TIME_GENERAL=1 MATINV_TIME_KERNEL=tilep timeout -k 10 300 python3 tools/time_sizes.py f64 gj 72 80 96 100 112 128 130 144 160 192 2>&1 | grep "n=" | tee -a $O/log.txt
done
for w in 4 blk; do
  echo "-- f32 MATINV_TILEP_WAVES=$w" | tee -a $O/log.txt
  MATINV_TILEP_WAVES=$w MATINV_TIME_GENERAL=1 MATINV_TIME_KERNEL=tilep timeout -k 10 300 python3 tools/time_sizes.py f32 gj 80 96 128 160 256 2>&1 | grep "n=" | tee -a $O/log.txt
done
echo "== multi_test 64 9000 3, eight times ==" | tee -a $O/log.txt
for i in 1 2 3 4 5 6 7 8; do (cd cuda-matrix-inversion_amd/host && timeout -k 10 120 ./multi_test 64 9000 3 2>&1 | tail -2) | tee -a $O/log.txt; done
echo "== full gpu tests ==" | tee -a $O/log.txt
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1
echo "rc=$?" | tee -a $O/log.txt; tail -15 $O/pytest_gpu.txt | tee -a $O/log.txt
