#!/bin/bash
# round 3, call 16: the whole library with -amdgpu-mfma-vgpr-form=1 against the default build: every bench workload + a few sweep sizes
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03p
mkdir -p $O
cd $R
for v in "" _vf; do
  L=$R/cuda-matrix-inversion_amd/libmatinv_hip$v.so
  MATINV_LIB=$L timeout -k 10 400 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench$v.json 2> $O/bench$v.err
  echo "== lib$v rc=$?"
  MATINV_LIB=$L timeout -k 10 100 python3 tools/time_sizes.py f64 chol 80 96 112 128 160 192 2>&1 | grep "n=" > $O/sizes$v.txt
  MATINV_LIB=$L timeout -k 10 100 python3 tools/time_sizes.py f32 gj 96 128 192 256 2>&1 | grep "n=" >> $O/sizes$v.txt
  MATINV_LIB=$L timeout -k 10 100 python3 tools/time_gp_sizes.py f64 64 80 96 112 128 2>&1 | grep "n=" >> $O/sizes$v.txt
done
python3 - <<'PY'
import json,os
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r03p'
a=json.load(open(O+'/bench.json')); b=json.load(open(O+'/bench_vf.json'))
print(f"headline {a['value']:.4g} -> {b['value']:.4g}")
for k in a['other_workloads']:
    x=a['other_workloads'][k]['inversions_per_s']; y=b['other_workloads'][k]['inversions_per_s']
    print(f"{k:22s} {x:10.4g} -> {y:10.4g}  {y/x:5.2f}")
print('mixed', a['mixed']['value'], b['mixed']['value'])
PY
paste -d'|' $O/sizes.txt $O/sizes_vf.txt | awk -F'|' '{split($1,a," "); split($2,b," "); print $1; print "   vf: " $2}' | cut -c1-150
