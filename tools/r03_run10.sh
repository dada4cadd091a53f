#!/bin/bash
# round 3, call 10: fp32 one-wave symmetric sweep 9 x 9 / 10 x 10 tiles -- parity of the new sizes, rates against the four-wave-wide kernels
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03j
mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "cholesky_large or pipeline_synthetic or not_spd or full_size" > $O/pytest.txt 2>&1
echo "rc=$?" | tee $O/log.txt; tail -3 $O/pytest.txt | tee -a $O/log.txt
for w in 1 0; do
  echo "== MATINV_ONEWAVE_WIDE=$w chol f32 ==" | tee -a $O/log.txt
  MATINV_ONEWAVE_WIDE=$w timeout -k 10 200 python3 tools/time_sizes.py f32 chol 128 130 144 150 160 161 176 2>&1 | grep "n=" | tee -a $O/log.txt
  echo "== MATINV_ONEWAVE_WIDE=$w pipeline f32 ==" | tee -a $O/log.txt
  MATINV_ONEWAVE_WIDE=$w timeout -k 10 200 python3 tools/time_gp_sizes.py f32 128 130 144 160 161 2>&1 | grep "n=" | tee -a $O/log.txt
done
