#!/bin/bash
# round 3, sixth GPU call: one-wavefront kernels on VGPRs + AGPRs (A/B against the several-wavefront kernels), tests of the new paths
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03f
mkdir -p $O
cd $R
echo "== correctness of the touched paths ==" | tee $O/log.txt
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "spd_fp64 or cholesky or pipeline or default_policy or tile_family or full_size" > $O/pytest_sub.txt 2>&1
echo "rc=$?" | tee -a $O/log.txt; tail -4 $O/pytest_sub.txt | tee -a $O/log.txt
for sw in 1 0; do
  echo "-- MATINV_ONEWAVE_WIDE=$sw: f64 gj (natural), chol, pipeline; f32 chol, pipeline" | tee -a $O/log.txt
  MATINV_ONEWAVE_WIDE=$sw timeout -k 10 200 python3 tools/time_sizes.py f64 gj 72 80 88 96 2>&1 | grep "n=" | tee -a $O/log.txt
  MATINV_ONEWAVE_WIDE=$sw timeout -k 10 200 python3 tools/time_sizes.py f64 chol 96 100 104 112 2>&1 | grep "n=" | tee -a $O/log.txt
  MATINV_ONEWAVE_WIDE=$sw timeout -k 10 200 python3 tools/time_gp_sizes.py f64 96 100 104 112 2>&1 | grep "n=" | tee -a $O/log.txt
  MATINV_ONEWAVE_WIDE=$sw timeout -k 10 200 python3 tools/time_sizes.py f32 chol 112 120 128 2>&1 | grep "n=" | tee -a $O/log.txt
  MATINV_ONEWAVE_WIDE=$sw timeout -k 10 200 python3 tools/time_gp_sizes.py f32 112 120 128 2>&1 | grep "n=" | tee -a $O/log.txt
done
echo "-- bordered one-wave pipeline kernel with AGPRs (MATINV_GP_SPD_TILE=0) at 84..96 f64" | tee -a $O/log.txt
MATINV_GP_SPD_TILE=0 timeout -k 10 200 python3 tools/time_gp_sizes.py f64 84 88 96 2>&1 | grep "n=" | tee -a $O/log.txt
