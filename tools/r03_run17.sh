#!/bin/bash
# round 3, call 17: SQ counters of the kernels added late in the round (wide fp32 / fp64 symmetric sweeps, three-wave pivoting kernel)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03q
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
: > $O/pmc_new_kernels.txt
run() {  # tag env... -- args of time_sizes.py
  tag=$1; shift
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS" "GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAVES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY"; do
    T=/tmp/pmcn_$$; rm -rf $T; mkdir -p $T
    timeout -k 10 120 rocprofv3 --kernel-trace --pmc $set -d $T -o a -- python3 $R/tools/time_sizes.py "$@" > $T/out.txt 2>&1
    python3 $R/tools/pmc_dump.py $T $T/d.txt "$tag" > /dev/null 2>&1
    grep -v "^#\|^pass\|worklist" $T/d.txt | sed "s/^a /$tag /" | cut -c1-170 >> $O/pmc_new_kernels.txt
  done
}
export MATINV_TIME_BATCH=8000
run chol_f32_144 f32 chol 144
run chol_f32_160 f32 chol 160
run chol_f64_112 f64 chol 112
MATINV_TIME_GENERAL=1 run gen_f64_96 f64 gj 96
MATINV_TIME_GENERAL=1 MATINV_TILEP_W3=0 run gen_f64_96_four_waves f64 gj 96
cat $O/pmc_new_kernels.txt
