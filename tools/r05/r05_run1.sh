#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_fused_splitk_gpu.py -x -q > gpurun_out/r05_a_fused_tests.log 2>&1; echo "fused tests rc=$?" | tee -a gpurun_out/r05_a_fused_tests.log
tail -25 gpurun_out/r05_a_fused_tests.log
timeout 1500 python -m pytest tests -m gpu -q -x --durations=8 > gpurun_out/r05_a_tests.log 2>&1; echo "suite rc=$?" | tee -a gpurun_out/r05_a_tests.log
tail -30 gpurun_out/r05_a_tests.log
tools/ab_bench.sh r05_a_fused CS_NO_FUSED_REDUCE 1 7
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_r05_a3 -o bench -- python $REPO/bench.py --no-cpu-baseline --no-traffic --no-extras --no-fp32-leg --objects 1 --steps 20 --warmup 3 > $REPO/gpurun_out/r05_a_bench_c2_under_rocprof.json 2> /dev/null
cd $REPO
DB=$(find gpurun_out/prof_r05_a3 -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB > gpurun_out/r05_a_c2_kernel_stats.txt
rm -rf gpurun_out/prof_r05_a3
head -30 gpurun_out/r05_a_c2_kernel_stats.txt | cut -c1-200
