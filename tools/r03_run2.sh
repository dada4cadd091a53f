#!/bin/bash
# round 3, second GPU call: the full -m gpu suite (multi-device host path, nccl group of one, C all-gather, per-matrix policy) + bench
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03b
mkdir -p $O
cd $R
echo "== full gpu tests ==" | tee $O/log.txt
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1
echo "rc=$?" | tee -a $O/log.txt; tail -15 $O/pytest_gpu.txt | tee -a $O/log.txt
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
    print("end_to_end", d.get("end_to_end"))
    m = d.get("mixed"); print("mixed", None if m is None else {k: m[k] for k in ("value", "ms_per_step", "host_ms_per_step", "host_share")})
except Exception as e:
    print("bench parse failed", e)
PY
echo "== end to end, 1 vs 2 vs 4 virtual shards on one device (host link is shared: expect no gain, only no loss) ==" | tee -a $O/log.txt
for nd in 1 2 4; do MATINV_DEVICES=$nd timeout -k 10 120 python3 tools/time_host_api.py 64 100000 2>&1 | grep inverse_gauss | sed "s/^/MATINV_DEVICES=$nd /" | tee -a $O/log.txt; done
