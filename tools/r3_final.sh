# round-3 final evidence run: suite, default bench line (+ live traffic), rocprofv3 summaries of the default command (whole
# process and per grid) and of the step loop alone, HBM-bound kernels, decode table
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( CS_PARITY_LOG=$PWD/gpurun_out/r03_z_parity_log.txt timeout 1500 python -m pytest tests -m gpu -q --durations=5 > gpurun_out/r03_z_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_z_tests.log )
tail -4 gpurun_out/r03_z_tests.log
timeout 900 python bench.py --traffic > gpurun_out/r03_z_bench.json 2> gpurun_out/r03_z_bench.err; echo "bench rc=$?"
python -c "import json; d=json.load(open('gpurun_out/r03_z_bench.json')); r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r['avg_launch_ms'], r['launches'], r['traffic'], d['c2']['ms_per_step'], d['c7']['ms_per_step'], d['decode']['ms_per_object'], d['cpu_baseline']['value'], d['fp32_mfma']['value'])"
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_r03z -o bench -- python $REPO/bench.py --no-cpu-baseline --no-traffic --gemm-table > $REPO/gpurun_out/r03_z_bench_under_rocprof.json 2> $REPO/gpurun_out/r03_z_gemm_table.txt
timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_r03z2 -o bench -- python $REPO/bench.py --no-cpu-baseline --no-extras > $REPO/gpurun_out/r03_z_bench_steploop_under_rocprof.json 2> /dev/null
cd $REPO
DB=$(find gpurun_out/prof_r03z -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB > gpurun_out/r03_z_kernel_stats.txt && python tools/rocpd_by_grid.py $DB "conv_gemm_f16x3_kernel<1, 7, 8, 1, true, 32," > gpurun_out/r03_z_dominant_kernel_by_grid.txt
DB=$(find gpurun_out/prof_r03z2 -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB > gpurun_out/r03_z_kernel_stats_steploop.txt && head -6 gpurun_out/r03_z_kernel_stats_steploop.txt
rm -rf gpurun_out/prof_r03z gpurun_out/prof_r03z2
python tools/hbm_bench.py > gpurun_out/r03_z_hbm_bound_kernels.txt 2>&1
python tools/decode_bench.py > gpurun_out/r03_z_decode_table.txt 2>&1
head -3 gpurun_out/r03_z_decode_table.txt
