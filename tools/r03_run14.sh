#!/bin/bash
# round 3, call 14: blocked Gauss-Jordan update with hoisted address arithmetic -- parity, rates, VALU per MFMA
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03n
mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "large_n or blocked or chunk" > $O/pytest.txt 2>&1
echo "rc=$?" | tee $O/log.txt; tail -3 $O/pytest.txt | tee -a $O/log.txt
MATINV_TIME_GENERAL=1 timeout -k 10 200 python3 tools/time_sizes.py f64 gj 200 224 256 320 384 512 768 1024 2>&1 | grep "n=" | tee -a $O/log.txt
MATINV_TIME_GENERAL=1 timeout -k 10 200 python3 tools/time_sizes.py f32 gj 300 512 1024 2>&1 | grep "n=" | tee -a $O/log.txt
bash tools/pmc_quick.sh gj1024g 256 auto 2>&1 | grep -E "update_mfma<double, false" | grep -E "INSTS_VALU |INSTS_MFMA|WAVE_CYCLES|WAIT_INST_ANY|GRBM|MFMA_BUSY" | tee -a $O/log.txt
