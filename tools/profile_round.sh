#!/bin/bash
# Collect the rocprofv3 evidence of one round on the GPU box and leave only compact summaries under gpurun_out/prof/.
# usage (inside gpurun): bash tools/profile_round.sh r01
# Kernel trace and PMC counters are taken in SEPARATE passes (gpurun refuses --pmc together with sys/runtime tracing).
set -u
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof
W=/tmp/matinv_prof_$$
mkdir -p $OUT $W
cd /tmp && export TMPDIR=/tmp

# the driver's command by itself first (under the profiler the host side of the mixed-size workload is several times slower), then traced
python3 $R/bench.py > $OUT/${TAG}_bench_default.json 2> $W/plain.err
rocprofv3 --kernel-trace --stats -d $W -o trace -- python3 $R/bench.py --steps 20 --warmup 3 > $OUT/${TAG}_bench_default_under_rocprofv3.json 2> $W/trace.err
python3 $R/tools/rocprof_summary.py $W/trace_results.db "$TAG: rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 3" > $OUT/${TAG}_bench_default_kernel_trace.txt

# the headline workload alone: the default line above also launches the headline kernel on other batch sizes (end-to-end chunks,
# residual checks), which pollutes its average; here every launch of it is one timed-size step
rocprofv3 --kernel-trace --stats -d $W -o headline -- python3 $R/bench.py --steps 20 --warmup 5 --no-others --no-cpu-baseline > $OUT/${TAG}_bench_headline.json 2> $W/headline.err
python3 $R/tools/rocprof_summary.py $W/headline_results.db "$TAG: rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 5 --no-others --no-cpu-baseline (the headline workload alone: every launch of its kernel is one 100 000-matrix step)" > $OUT/${TAG}_bench_headline_kernel_trace.txt

for w in gj64 gj16 gj24 gj32 chol64 gj128 gj64g gj32g gj128g gj192g chol144; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c -d $W -o ${w}_$c -- python3 $R/bench.py --workload $w --steps 3 --warmup 2 --no-cpu-baseline --no-others > $W/${w}_$c.out 2>&1
  done
done
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $W -o gj64_SQ1 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-others > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE -d $W -o gj64_SQ2 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-others > /dev/null 2>&1
for w in gj64g gj128g gj192g chol144 chol1024 gj1024g gj256g; do
B=""; [ $w = chol1024 ] && B="--batch 256"; [ $w = gj1024g ] && B="--batch 256"; [ $w = gj256g ] && B="--batch 3000"; [ $w = gj192g ] && B="--batch 5000"; [ $w = chol144 ] && B="--batch 10000"; [ $w = gj128g ] && B="--batch 25000"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $W -o ${w}_SQ1 -- python3 $R/bench.py --workload $w $B --steps 3 --warmup 2 --no-cpu-baseline --no-others > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_SALU -d $W -o ${w}_SQ2 -- python3 $R/bench.py --workload $w $B --steps 3 --warmup 2 --no-cpu-baseline --no-others > /dev/null 2>&1
done
# the headline kernel's memory side: the same counters on the load+store-only build (tools/build_ldst_variant.sh) and on the real one
rocprofv3 -L > $OUT/${TAG}_counter_list.txt 2>&1
MEM1="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"
MEM2="TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum"
MEM3="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum"
MEM4="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE"
i=0
for C in "$MEM1" "$MEM2" "$MEM3" "$MEM4"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C -d $W -o gj64_MEM$i -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-others > $W/mem$i.out 2>&1
  if [ -f $R/cuda-matrix-inversion_amd/libmatinv_hip_ldst.so ]; then
    MATINV_LIB=$R/cuda-matrix-inversion_amd/libmatinv_hip_ldst.so MATINV_BENCH_NO_RESIDUAL=1 rocprofv3 --kernel-trace --pmc $C -d $W -o gj64ldst_MEM$i -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-others > $W/memldst$i.out 2>&1
  fi
done
if [ -f $R/cuda-matrix-inversion_amd/libmatinv_hip_ldst.so ]; then
  MATINV_LIB=$R/cuda-matrix-inversion_amd/libmatinv_hip_ldst.so MATINV_BENCH_NO_RESIDUAL=1 rocprofv3 --kernel-trace --stats -d $W -o ldst_trace -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-others > $W/ldst_trace.out 2>&1
  python3 $R/tools/rocprof_summary.py $W/ldst_trace_results.db "$TAG: load+store-only build of the natural-order tile kernels (MATINV_LIB=libmatinv_hip_ldst.so): rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 3" > $OUT/${TAG}_ldst_only_kernel_trace.txt
fi
python3 $R/tools/pmc_dump.py $W $OUT/${TAG}_pmc_counters.txt "$TAG"
rm -rf $W
ls -la $OUT
