cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/r04_g_tests.log; tail -25 gpurun_out/r04_g_tests.log
for mode in on off; do
  if [ $mode = off ]; then export CS_NO_GN_PARTS=1 CS_NO_PAIR_EPILOGUE=1 CS_NO_DYN_SCALE=1; fi
  timeout 600 python bench.py --no-cpu-baseline --no-traffic --no-fp32-leg > gpurun_out/r04_g_bench_$mode.json 2> gpurun_out/r04_g_bench_$mode.err
  python - <<PY
import json
try:
    d = json.load(open('gpurun_out/r04_g_bench_$mode.json')); r = d['roofline']
    print('$mode: steps/s', d['value'], 'ms', d['ms_per_step'], 'frac', r['frac'], 'c2', d['c2']['ms_per_step'], 'c7', d['c7']['ms_per_step'], 'dec', d['decode']['ms_per_object'], 'c7x5', d['c7x5']['default_api']['seconds'], 'native', d.get('native_driver'))
except Exception as e:
    print('no bench line', e)
PY
  tail -2 gpurun_out/r04_g_bench_$mode.err
done
