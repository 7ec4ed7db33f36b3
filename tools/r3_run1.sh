# round-3 GPU run 1: the whole -m gpu suite (incl. the new full-size C4 / C5 tests), the default bench line, a rocprofv3
# kernel summary of the same command, and the split4-with-pre-split A/B (cs_gemm.hip: split4_large)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( CS_PARITY_LOG=$PWD/gpurun_out/r03_a_parity_log.txt timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/r03_a_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_a_tests.log )
tail -5 gpurun_out/r03_a_tests.log
timeout 600 python bench.py > gpurun_out/r03_a_bench.json 2> gpurun_out/r03_a_bench.err; echo "bench rc=$?"
tail -c 600 gpurun_out/r03_a_bench.err
for flag in "-DCS_SPLIT4_FP32_ONLY" ""; do
  CS_EXTRA_HIPCC_FLAGS="$flag" python -m commonscenes_amd.build --force > /dev/null 2>&1
  for rep in 1 2; do
    timeout 300 python bench.py --no-cpu-baseline --no-fp32-leg --steps 10 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('split4[$flag]', 'ms/step', round(d['ms_per_step'],2), 'frac', round(d['roofline']['frac'],4), 'decode ms', round(d['decode']['ms'],1), 'c2', round(d['c2']['ms_per_step'],2), 'c7', round(d['c7']['ms_per_step'],2))"
  done
done | tee gpurun_out/r03_a_split4_ab.txt
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_r03a -o bench -- python $REPO/bench.py --no-cpu-baseline --no-fp32-leg --gemm-table > $REPO/gpurun_out/r03_a_bench_under_rocprof.json 2> $REPO/gpurun_out/r03_a_gemm_table.txt
cd $REPO
DB=$(find gpurun_out/prof_r03a -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB > gpurun_out/r03_a_kernel_stats.txt && head -30 gpurun_out/r03_a_kernel_stats.txt
