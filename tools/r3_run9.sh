# round-3 GPU run 9: the suite, the default bench line (+ live traffic), the rocprofv3 summary of the same command
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( CS_PARITY_LOG=$PWD/gpurun_out/r03_i_parity_log.txt timeout 1500 python -m pytest tests -m gpu -q --durations=5 > gpurun_out/r03_i_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_i_tests.log )
tail -10 gpurun_out/r03_i_tests.log
grep -h "tiny-gamma\|heavy-tailed\|uniformly\|full-size <\|C5 full" gpurun_out/r03_i_tests.log | sort -u | head
timeout 900 python bench.py --traffic > gpurun_out/r03_i_bench.json 2> gpurun_out/r03_i_bench.err; echo "bench rc=$?"
python -c "import json; d=json.load(open('gpurun_out/r03_i_bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['c2']['ms_per_step'], d['c7']['ms_per_step'], d['decode']['ms_per_object'], d['cpu_baseline']['value'], d['fp32_mfma']['value'])"
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_r03i -o bench -- python $REPO/bench.py --no-cpu-baseline --no-fp32-leg --gemm-table > $REPO/gpurun_out/r03_i_bench_under_rocprof.json 2> $REPO/gpurun_out/r03_i_gemm_table.txt
cd $REPO
DB=$(find gpurun_out/prof_r03i -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB > gpurun_out/r03_i_kernel_stats.txt && head -8 gpurun_out/r03_i_kernel_stats.txt
rm -rf gpurun_out/prof_r03i
python tools/hbm_bench.py > gpurun_out/r03_i_hbm_bound_kernels.txt 2>&1; tail -20 gpurun_out/r03_i_hbm_bound_kernels.txt
