#!/bin/bash
# round 3, call 12: reject-count test over all kernel families; fp32 9 x 9 / 10 x 10 one-wave sweep at two waves per SIMD (spills) against tile4-wide
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03l
mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu  > $O/pytest.txt 2>&1
echo "rc=$?" | tee $O/log.txt; tail -30 $O/pytest.txt | tee -a $O/log.txt
for w in 1 0; do
  echo "== MATINV_ONEWAVE_WIDE=$w chol f32 ==" | tee -a $O/log.txt
  MATINV_ONEWAVE_WIDE=$w timeout -k 10 200 python3 tools/time_sizes.py f32 chol 130 144 150 160 2>&1 | grep "n=" | tee -a $O/log.txt
  echo "== MATINV_ONEWAVE_WIDE=$w pipeline f32 ==" | tee -a $O/log.txt
  MATINV_ONEWAVE_WIDE=$w timeout -k 10 200 python3 tools/time_gp_sizes.py f32 130 144 160 2>&1 | grep "n=" | tee -a $O/log.txt
done
