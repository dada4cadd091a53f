#!/bin/bash
# round 3, call 11: why is the 9 x 9-tile fp32 one-wave sweep 15x slower than its MFMA count says? instruction-fetch counters
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03k
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 -L > $O/counters.txt 2>&1
grep -io "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQC_INST[A-Z_]*" $O/counters.txt | sort -u | tee $O/ic_names.txt
for n in 128 144; do
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_IFETCH SQ_IFETCH_LEVEL" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_WAIT_ANY SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS"; do
    T=/tmp/pmck_$$_$n; rm -rf $T; mkdir -p $T
    MATINV_TIME_BATCH=4000 timeout -k 10 120 rocprofv3 --kernel-trace --pmc $set -d $T -o a -- python3 $R/tools/time_sizes.py f32 chol $n > $T/out.txt 2>&1
    echo "== n=$n set: $set rc=$?" >> $O/pmc.txt
    python3 $R/tools/pmc_dump.py $T $T/d.txt quick > /dev/null 2>&1; grep -v worklist $T/d.txt | cut -c1-200 >> $O/pmc.txt
    tail -2 $T/out.txt >> $O/pmc.txt
  done
done
cat $O/pmc.txt
