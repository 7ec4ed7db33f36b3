# round-3 GPU run 3: suite, default bench line (+ --traffic), rocprofv3 summary, GEMM tables at 7 objects and 1 object
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( CS_PARITY_LOG=$PWD/gpurun_out/r03_c_parity_log.txt timeout 1500 python -m pytest tests -m gpu -q --durations=5 > gpurun_out/r03_c_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_c_tests.log )
tail -12 gpurun_out/r03_c_tests.log
timeout 900 python bench.py --traffic > gpurun_out/r03_c_bench.json 2> gpurun_out/r03_c_bench.err; echo "bench rc=$?"
python -c "import json; d=json.load(open('gpurun_out/r03_c_bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['c2']['ms_per_step'], d['c7']['ms_per_step'], d['decode']['ms_per_object'], d['cpu_baseline']['value'])"
for nobj in 7 1; do
  timeout 300 python bench.py --no-cpu-baseline --no-fp32-leg --no-extras --objects $nobj --steps 10 --warmup 3 --gemm-table > gpurun_out/r03_c_bench_obj$nobj.json 2> gpurun_out/r03_c_gemm_table_obj$nobj.txt
done
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_r03c -o bench -- python $REPO/bench.py --no-cpu-baseline --no-fp32-leg --gemm-table > $REPO/gpurun_out/r03_c_bench_under_rocprof.json 2> $REPO/gpurun_out/r03_c_gemm_table.txt
timeout 300 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_r03c7 -o bench -- python $REPO/bench.py --no-cpu-baseline --no-fp32-leg --no-extras --objects 7 --steps 10 --warmup 3 > /dev/null 2>&1
cd $REPO
DB=$(find gpurun_out/prof_r03c -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB > gpurun_out/r03_c_kernel_stats.txt && head -12 gpurun_out/r03_c_kernel_stats.txt
DB=$(find gpurun_out/prof_r03c7 -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB > gpurun_out/r03_c_kernel_stats_obj7.txt && head -25 gpurun_out/r03_c_kernel_stats_obj7.txt
rm -rf gpurun_out/prof_r03c gpurun_out/prof_r03c7
