#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=r05_b
timeout 900 python -m pytest tests/test_fused_splitk_gpu.py tests/test_vae_facade_gpu.py tests/test_epilogue_outputs_gpu.py -x -q > gpurun_out/${T}_new_tests.log 2>&1; echo "new tests rc=$?" | tee -a gpurun_out/${T}_new_tests.log
tail -30 gpurun_out/${T}_new_tests.log
timeout 1800 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/${T}_tests.log 2>&1; echo "suite rc=$?" | tee -a gpurun_out/${T}_tests.log
tail -30 gpurun_out/${T}_tests.log
tools/ab_bench.sh ${T}_fused CS_NO_FUSED_REDUCE 1 7
tools/ab_bench.sh ${T}_up2 CS_NO_UP2_BATCH 1
tools/ab_bench.sh ${T}_gnfold CS_NO_GN_FOLD 1 7
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_${T}3 -o bench -- python $REPO/bench.py --no-cpu-baseline --no-traffic --no-extras --no-fp32-leg --objects 1 --steps 20 --warmup 3 > $REPO/gpurun_out/${T}_bench_c2_under_rocprof.json 2> /dev/null
cd $REPO
DB=$(find gpurun_out/prof_${T}3 -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB > gpurun_out/${T}_c2_kernel_stats.txt
rm -rf gpurun_out/prof_${T}3
head -32 gpurun_out/${T}_c2_kernel_stats.txt | cut -c1-200
timeout 900 python tools/cpu_thread_sweep.py > gpurun_out/${T}_cpu_threads.txt 2>&1
cat gpurun_out/${T}_cpu_threads.txt | tail -8
