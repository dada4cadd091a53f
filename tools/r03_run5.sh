#!/bin/bash
# round 3, fifth GPU call: full suite on the rebuilt library, default bench line, fused pipeline on the SPD sweep (A/B)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03e
mkdir -p $O
cd $R
echo "== full gpu tests ==" | tee $O/log.txt
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1
echo "rc=$?" | tee -a $O/log.txt; tail -12 $O/pytest_gpu.txt | tee -a $O/log.txt
echo "== fused pipeline: SPD-sweep kernel on/off ==" | tee -a $O/log.txt
for sw in 1 0; do
  echo "-- MATINV_GP_SPD_TILE=$sw" | tee -a $O/log.txt
  MATINV_GP_SPD_TILE=$sw timeout -k 10 200 python3 tools/time_gp_sizes.py f64 80 84 88 96 100 2>&1 | grep "n=" | tee -a $O/log.txt
  MATINV_GP_SPD_TILE=$sw timeout -k 10 200 python3 tools/time_gp_sizes.py f32 96 100 104 112 120 2>&1 | grep "n=" | tee -a $O/log.txt
done
echo "== bench default ==" | tee -a $O/log.txt
timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
echo "rc=$?" | tee -a $O/log.txt; tail -c 600 $O/bench_default.err | tee -a $O/log.txt
python3 - <<PY | tee -a $O/log.txt
import json
try:
    d = json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
    print("value", d["value"], "frac", d["roofline"]["frac"])
    for k, v in d.get("other_workloads", {}).items():
        print(f"  {k:22s} {v['inversions_per_s']:.3e} inv/s {v['bound']} {v['frac']:.3f} resid {v['residual_max_64']:.1e} {v['kernel']}")
    print("end_to_end", {k: d["end_to_end"][k] for k in ("ms", "inversions_per_s", "host_link_GBs_both_directions")})
    m = d.get("mixed"); print("mixed", None if m is None else {k: m[k] for k in ("value", "ms_per_step", "host_ms_per_step", "host_share", "one_flush_at_a_time")})
except Exception as e:
    print("bench parse failed", e)
PY
