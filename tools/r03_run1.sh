#!/bin/bash
# round 3, first GPU call: new per-column pivoting kernel at 64 < n <= 128 (A/B), bits of natural-new vs pivot, tests, bench
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03a
mkdir -p $O
cd $R
echo "== tilepw NT 5..8 correctness (general, singular, fixtures) ==" | tee $O/log.txt
MATINV_TILEP_WAVES=col timeout -k 10 400 python3 -m pytest tests/test_gpu_parity.py -x -q -k "general or pivot or tilep or singular or square" > $O/pytest_col.txt 2>&1
echo "rc=$?" | tee -a $O/log.txt; tail -3 $O/pytest_col.txt | tee -a $O/log.txt
echo "== A/B general f64 (tilep kernel forced) ==" | tee -a $O/log.txt
for w in 4 col; do
  echo "-- MATINV_TILEP_WAVES=$w" | tee -a $O/log.txt
  MATINV_TILEP_WAVES=$w MATINV_TIME_GENERAL=1 MATINV_TIME_KERNEL=tilep timeout -k 10 300 python3 tools/time_sizes.py f64 gj 72 80 96 100 112 128 2>&1 | grep "n=" | tee -a $O/log.txt
done
for w in 4 col; do
  echo "-- f32 MATINV_TILEP_WAVES=$w" | tee -a $O/log.txt
  MATINV_TILEP_WAVES=$w MATINV_TIME_GENERAL=1 MATINV_TIME_KERNEL=tilep timeout -k 10 300 python3 tools/time_sizes.py f32 gj 80 96 128 2>&1 | grep "n=" | tee -a $O/log.txt
done
echo "== bits: natural (old/new) vs pivot on SPD ==" | tee -a $O/log.txt
for n in 32 64; do
  MATINV_GJ_POLICY=natural MATINV_TILE_NATURAL=old python3 tools/ab_bits.py $n f64 spd 2>&1 | grep sha | tee -a $O/log.txt
  MATINV_GJ_POLICY=natural MATINV_TILE_NATURAL=new python3 tools/ab_bits.py $n f64 spd 2>&1 | grep sha | tee -a $O/log.txt
  MATINV_GJ_POLICY=pivot python3 tools/ab_bits.py $n f64 spd 2>&1 | grep sha | tee -a $O/log.txt
done
echo "== full gpu tests ==" | tee -a $O/log.txt
timeout -k 10 600 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1
echo "rc=$?" | tee -a $O/log.txt; tail -5 $O/pytest_gpu.txt | tee -a $O/log.txt
echo "== bench default ==" | tee -a $O/log.txt
timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
echo "rc=$?" | tee -a $O/log.txt; tail -c 600 $O/bench_default.err | tee -a $O/log.txt
python3 - <<PY | tee -a $O/log.txt
import json
try:
    d = json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
    print("value", d["value"], "frac", d["roofline"]["frac"])
    for k, v in d.get("other_workloads", {}).items():
        print(f"  {k:10s} {v['inversions_per_s']:.3e} inv/s {v['bound']} {v['frac']:.3f} resid {v['residual_max_64']:.1e}")
    print("end_to_end", d.get("end_to_end"))
    m = d.get("mixed"); print("mixed", None if m is None else {k: m[k] for k in ("value", "ms_per_step", "host_ms_per_step", "host_share")})
except Exception as e:
    print("bench parse failed", e)
PY
