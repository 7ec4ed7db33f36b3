cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_epilogue_outputs_gpu.py tests/test_mesh.py -q 2>&1 | tail -30 > gpurun_out/r04_d_epi_tests.log; tail -14 gpurun_out/r04_d_epi_tests.log
timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_epilogue_outputs_gpu.py --deselect tests/test_mesh.py 2>&1 | tail -25 > gpurun_out/r04_d_tests.log; tail -4 gpurun_out/r04_d_tests.log
for mode in on off; do
  if [ $mode = off ]; then export CS_NO_GN_PARTS=1 CS_NO_PAIR_EPILOGUE=1; fi
  timeout 600 python bench.py --no-cpu-baseline --no-traffic --no-fp32-leg > gpurun_out/r04_d_bench_$mode.json 2> gpurun_out/r04_d_bench_$mode.err
  python - <<PY
import json
try:
    d = json.load(open('gpurun_out/r04_d_bench_$mode.json')); r = d['roofline']
    print('$mode: steps/s', d['value'], 'ms', d['ms_per_step'], 'frac', r['frac'], 'c2', d['c2']['ms_per_step'], 'c7', d['c7']['ms_per_step'], 'dec', d['decode']['ms_per_object'], 'c7x5', d['c7x5']['default_api']['seconds'])
except Exception as e:
    print('no bench line', e)
PY
done
unset CS_NO_GN_PARTS CS_NO_PAIR_EPILOGUE
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_d -o bench -- python $REPO/bench.py --no-cpu-baseline --no-extras > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_d3 -o bench -- python $REPO/bench.py --no-cpu-baseline --no-extras --objects 1 --steps 20 --warmup 3 > /dev/null 2>&1
cd $REPO
DB=$(find gpurun_out/prof_d -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_summary.py $DB > gpurun_out/r04_d_kernel_stats_steploop.txt && head -30 gpurun_out/r04_d_kernel_stats_steploop.txt | cut -c1-170
DB=$(find gpurun_out/prof_d3 -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_summary.py $DB > gpurun_out/r04_d_c2_kernel_stats.txt && head -30 gpurun_out/r04_d_c2_kernel_stats.txt | cut -c1-170
rm -rf gpurun_out/prof_d gpurun_out/prof_d3
