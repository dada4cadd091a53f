#!/bin/bash
# The reference's own benchmark protocol (README.md:7-10 there: log=1, BENCH_REPS=10, 100 matrices x DUPS=16 = 1600, fp32,
# 8 OpenMP threads) on synthetic fixtures of its sweep sizes: median of the 10 samples per timer key, in ms.
# usage (inside gpurun): bash tools/reference_keys.sh > gpurun_out/reference_keys.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
H=$R/cuda-matrix-inversion_amd/host
W=/tmp/matinv_refkeys_$$
mkdir -p $W
echo "# key,batch,n -> median ms over 10 reps (MATINV_DETAILED_LOGGING=1, OMP_NUM_THREADS=8, fp32 CLIs, 100 synthetic matrices x 16)"
for n in 8 16 32 64 128; do
  python3 $R/tools/generate_fixtures.py inverse $W/inv_$n 100 $n >/dev/null
  python3 $R/tools/generate_fixtures.py gaussian $W/gp_$n 100 $n >/dev/null
  MATINV_DETAILED_LOGGING=1 OMP_NUM_THREADS=8 $H/inverse_bench_f32 $W/inv_$n 10 16 2>/dev/null > $W/inv_$n.log
  MATINV_DETAILED_LOGGING=1 OMP_NUM_THREADS=8 $H/gauss_bench_f32 $W/gp_$n 10 16 2>/dev/null > $W/gp_$n.log
  python3 - $W/inv_$n.log $W/gp_$n.log <<'PY'
import sys, statistics, collections
for path in sys.argv[1:]:
    d = collections.OrderedDict()
    for ln in open(path):
        f = ln.strip().split(",")
        if len(f) == 5:
            try: d.setdefault((f[0], f[1], f[2]), []).append(float(f[3]))
            except ValueError: pass
    for (k, b, n), v in d.items():
        print(f"{k},{b},{n}  {statistics.median(v):.4f}  (n={len(v)})")
PY
done
rm -rf $W
