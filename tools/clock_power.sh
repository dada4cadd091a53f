#!/bin/bash
# Clock and power of the card while (a) the headline kernel and (b) its load+store-only build run, sampled from a separate process
# (tools/clock_sample.py) -> gpurun_out/prof/<tag>_clock_power_{full,ldst}.csv + a summary. usage (inside gpurun): bash tools/clock_power.sh r04
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd $R
for v in full ldst; do
  python3 tools/clock_sample.py $OUT/${TAG}_clock_power_$v.csv 25 &
  SPID=$!
  sleep 1.0
  if [ $v = ldst ]; then export MATINV_LIB=$R/cuda-matrix-inversion_amd/libmatinv_hip_ldst.so MATINV_BENCH_NO_RESIDUAL=1; fi
  timeout -k 10 300 python3 bench.py --steps 3000 --warmup 50 --no-others --no-cpu-baseline > $OUT/${TAG}_clock_power_$v.json 2> $OUT/${TAG}_clock_power_$v.err
  unset MATINV_LIB MATINV_BENCH_NO_RESIDUAL
  sleep 0.5
  kill $SPID; wait $SPID 2>/dev/null
done
python3 - $OUT $TAG <<'PY'
import sys, json, statistics
out, tag = sys.argv[1], sys.argv[2]
for v in ("full", "ldst"):
    lines = open(f"{out}/{tag}_clock_power_{v}.csv").read().splitlines()
    names = [ln for ln in lines if ln.startswith("# t_s")][0][2:].split(",")
    rows = [ln.split(",") for ln in lines if not ln.startswith("#")]
    try:
        j = json.loads([l for l in open(f"{out}/{tag}_clock_power_{v}.json") if l.startswith("{")][-1])
        ms, kms = j["ms_per_step"], j["roofline"]["kernel_ms"]
    except Exception:
        ms = kms = None
    def col(i):
        return [float(r[i]) if len(r) > i and r[i] else None for r in rows]
    pcols = [i for i, nm in enumerate(names) if nm.endswith("_power_W")]
    # the card under test: the one whose power reading MOVES most (a GPU box shows every card of its host, other tenants' included)
    def swing(i):
        xs = [x for x in col(i) if x is not None]
        return (max(xs) - min(xs)) if xs else 0
    best = max(pcols, key=swing)
    pw, clk = col(best), col(best - 1)
    idle = min(x for x in pw if x is not None)
    peak = max(x for x in pw if x is not None)
    busy = [k for k, x in enumerate(pw) if x is not None and x > idle + 0.8 * (peak - idle)]
    bc = [clk[k] for k in busy if clk[k] is not None]
    print(f"{v}: {names[best][:-8]}: ms_per_step {ms}, kernel_ms {kms}; {len(rows)} samples, {len(busy)} under load: "
          f"sclk median {statistics.median(bc):.0f} MHz (min {min(bc):.0f}, max {max(bc):.0f}), power median {statistics.median([pw[k] for k in busy]):.0f} W "
          f"(peak {peak:.0f} W, idle {idle:.0f} W)")
PY
