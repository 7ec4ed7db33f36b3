# round-3 GPU run 6: which of the two 512-row tiles pays (decode), the unsplit skip_connection, suite
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for arm in 1 9 8 ""; do
  for rep in 1 2; do
    CS_NO_TILE512=$arm timeout 300 python bench.py --no-cpu-baseline --no-fp32-leg --steps 6 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['decode']['roofline']; print('no512[$arm]', 'decode ms', round(d['decode']['ms'],2), 'ms/object', round(d['decode']['ms_per_object'],3), 'dominant', r['kernel'][23:48], round(r['achieved'],1), 'TF/s', 'all gemm', round(r['all_gemm_tflops'],1), 'step', round(d['ms_per_step'],2), 'c7', round(d['c7']['ms_per_step'],2))"
  done
done | tee gpurun_out/r03_f_tile512_ab.txt
( timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r03_f_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_f_tests.log )
tail -5 gpurun_out/r03_f_tests.log
