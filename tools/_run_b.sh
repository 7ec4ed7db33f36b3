cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_epilogue_outputs_gpu.py -q -x 2>&1 | tail -25 > gpurun_out/r04_b_epi_tests.log; tail -12 gpurun_out/r04_b_epi_tests.log
timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_epilogue_outputs_gpu.py 2>&1 | tail -25 > gpurun_out/r04_b_tests.log; tail -8 gpurun_out/r04_b_tests.log
timeout 600 python bench.py --no-cpu-baseline --no-traffic > gpurun_out/r04_b_bench.json 2> gpurun_out/r04_b_bench.err; echo "bench rc=$?"; tail -2 gpurun_out/r04_b_bench.err
python - <<'PY'
import json
try:
    d = json.load(open('gpurun_out/r04_b_bench.json')); r = d['roofline']
    print('steps/s', d['value'], 'ms', d['ms_per_step'], 'frac', r['frac'], 'c2', d['c2']['ms_per_step'], 'c7', d['c7']['ms_per_step'], 'dec', d['decode']['ms_per_object'], 'c7x5', d['c7x5']['default_api']['seconds'])
except Exception as e:
    print('no bench line', e)
PY
CS_NO_GN_PARTS=1 CS_NO_PAIR_EPILOGUE=1 timeout 600 python bench.py --no-cpu-baseline --no-traffic --no-fp32-leg > gpurun_out/r04_b_bench_off.json 2>/dev/null
python - <<'PY'
import json
try:
    d = json.load(open('gpurun_out/r04_b_bench_off.json'))
    print('OFF: steps/s', d['value'], 'ms', d['ms_per_step'], 'c2', d['c2']['ms_per_step'], 'c7', d['c7']['ms_per_step'])
except Exception as e:
    print('no bench line', e)
PY
SM_BATCH=2 timeout 600 python tools/gemm_smallm_tiles.py > gpurun_out/r04_b_smallm_tiles_b2.txt 2>&1; cat gpurun_out/r04_b_smallm_tiles_b2.txt | cut -c1-400
