# round-3 GPU run 18: pointwise specialisation of the gather kernel: tests, A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r03_p_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_p_tests.log )
tail -5 gpurun_out/r03_p_tests.log
for arm in 1 ""; do
  for rep in 1 2; do
    CS_NO_PW=$arm timeout 300 python bench.py --no-cpu-baseline --no-fp32-leg --steps 10 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nopw[$arm]', 'ms/step', round(d['ms_per_step'],2), 'decode ms', round(d['decode']['ms'],2), 'c2', round(d['c2']['ms_per_step'],2), 'c7', round(d['c7']['ms_per_step'],2))"
  done
done | tee gpurun_out/r03_p_pw_ab.txt
CS_NO_PW=1 timeout 300 python bench.py --no-cpu-baseline --no-fp32-leg --no-extras --gemm-table 2>&1 >/dev/null | grep "^ *1 " | head -9
timeout 300 python bench.py --no-cpu-baseline --no-fp32-leg --no-extras --gemm-table 2>&1 >/dev/null | grep "^ *1 " | head -9
