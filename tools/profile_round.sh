#!/bin/bash
# Collect the rocprofv3 evidence of one round on the GPU box and leave only compact summaries under gpurun_out/prof/.
# usage (inside gpurun): bash tools/profile_round.sh r01
# Kernel trace and PMC counters are taken in SEPARATE passes (gpurun refuses --pmc together with sys/runtime tracing).
set -u
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof
W=/tmp/matinv_prof_$$
mkdir -p $OUT $W
cd /tmp && export TMPDIR=/tmp

rocprofv3 --kernel-trace --stats -d $W -o trace -- python3 $R/bench.py --steps 20 --warmup 3 > $OUT/${TAG}_bench_default.json 2> $W/trace.err
python3 $R/tools/rocprof_summary.py $W/trace_results.db "$TAG: rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 3" > $OUT/${TAG}_bench_default_kernel_trace.txt

for w in gj64 gj16 gj24 chol64 gj128 gj64g gj32g gj128g; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c -d $W -o ${w}_$c -- python3 $R/bench.py --workload $w --steps 3 --warmup 2 --no-cpu-baseline --no-others > $W/${w}_$c.out 2>&1
  done
done
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $W -o gj64_SQ1 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-others > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE -d $W -o gj64_SQ2 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-others > /dev/null 2>&1
for w in gj64g gj128g chol1024; do
B=""; [ $w = chol1024 ] && B="--batch 256"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $W -o ${w}_SQ1 -- python3 $R/bench.py --workload $w $B --steps 3 --warmup 2 --no-cpu-baseline --no-others > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_SALU -d $W -o ${w}_SQ2 -- python3 $R/bench.py --workload $w $B --steps 3 --warmup 2 --no-cpu-baseline --no-others > /dev/null 2>&1
done
python3 $R/tools/pmc_dump.py $W $OUT/${TAG}_pmc_counters.txt "$TAG"
rm -rf $W
ls -la $OUT
