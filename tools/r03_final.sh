#!/bin/bash
# round 3: the closing GPU call -- full suite, default bench line, rocprofv3 evidence (tools/profile_round.sh), size sweep
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03z
mkdir -p $O
cd $R
echo "== full gpu tests ==" | tee $O/log.txt
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1
echo "rc=$?" | tee -a $O/log.txt; tail -4 $O/pytest_gpu.txt | tee -a $O/log.txt
echo "== pipeline 17..25 ==" | tee -a $O/log.txt
timeout -k 10 200 python3 tools/time_gp_sizes.py f64 17 20 24 25 2>&1 | grep "n=" | tee -a $O/log.txt
timeout -k 10 200 python3 tools/time_gp_sizes.py f32 20 24 2>&1 | grep "n=" | tee -a $O/log.txt
echo "== bench default ==" | tee -a $O/log.txt
timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
echo "rc=$?" | tee -a $O/log.txt
echo "== smoke ==" | tee -a $O/log.txt
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee -a $O/log.txt
echo "== profile round ==" | tee -a $O/log.txt
bash tools/profile_round.sh r03 > $O/profile_round.log 2>&1; tail -2 $O/profile_round.log | tee -a $O/log.txt
echo "== size sweep ==" | tee -a $O/log.txt
bash tools/size_sweep.sh > $R/gpurun_out/prof/r03_size_sweep.txt 2> $O/size_sweep.err; wc -l $R/gpurun_out/prof/r03_size_sweep.txt | tee -a $O/log.txt
