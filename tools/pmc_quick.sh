#!/bin/bash
# quick SQ counter snapshot of one bench workload: bash tools/pmc_quick.sh <workload> [batch] [kernel]
W=${1:-gj64}; B=${2:-100000}; K=${3:-auto}
R=${GRAFT_REPO_ROOT:-$(pwd)}; T=/tmp/pmcq_$$; mkdir -p $T; cd /tmp; export TMPDIR=/tmp
A="--workload $W --batch $B --kernel $K --steps 3 --warmup 1 --no-cpu-baseline --no-others"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $T -o a -- python3 $R/bench.py $A > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_SALU -d $T -o b -- python3 $R/bench.py $A > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CU_CYCLES SQ_INSTS_WAVE32_VALU -d $T -o c -- python3 $R/bench.py $A > /dev/null 2>&1
python3 $R/tools/pmc_dump.py $T /tmp/pmcq.txt quick > /dev/null; grep -v worklist /tmp/pmcq.txt | cut -c1-120
rm -rf $T
