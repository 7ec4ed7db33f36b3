#!/bin/bash
# kw epilogue-prefetch check: K-wave tests, clean 1 / 7-object timings, one-object GEMM table + kernel stats
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=r05_k
timeout 900 python -m pytest tests/test_kwave_gpu.py tests/test_fused_splitk_gpu.py -x -q 2>&1 | tail -3
for o in 1 7; do
for rep in 1 2; do
    ms=$(timeout 600 python bench.py --no-cpu-baseline --no-traffic --no-extras --no-fp32-leg --no-gemm-profile --objects $o --steps 40 --warmup 5 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "objects=$o kw_prefetch=1 rep=$rep ms_per_step=$ms" | tee -a gpurun_out/${T}_kw_prefetch.txt
done
done
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_${T}4 -o bench -- python $REPO/bench.py --no-cpu-baseline --no-traffic --no-extras --no-fp32-leg --objects 1 --steps 10 --warmup 3 --gemm-table > $REPO/gpurun_out/${T}_bench_c1_under_rocprof.json 2> $REPO/gpurun_out/${T}_gemm_table_c1.txt
cd $REPO
DB=$(find gpurun_out/prof_${T}4 -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB > gpurun_out/${T}_c1_kernel_stats.txt
rm -rf gpurun_out/prof_${T}4
head -30 gpurun_out/${T}_c1_kernel_stats.txt | cut -c1-180
grep -v "^[WEI]2026" gpurun_out/${T}_gemm_table_c1.txt | head -50
