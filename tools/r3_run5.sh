# round-3 GPU run 5: 512-row slab tiles (decode): tests, then A/B CS_NO_TILE512=1 vs default on the decode block
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1200 python -m pytest tests/test_f16x3_gpu.py tests/test_vqvae_native_gpu.py tests/test_model_gpu.py tests/test_ops_gpu.py -m gpu -q -x > gpurun_out/r03_e_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_e_tests.log )
tail -8 gpurun_out/r03_e_tests.log
for arm in 1 ""; do
  for rep in 1 2; do
    CS_NO_TILE512=$arm timeout 300 python bench.py --no-cpu-baseline --no-fp32-leg --steps 4 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['decode']['roofline']; print('no512[$arm]', 'decode ms', round(d['decode']['ms'],2), 'ms/object', round(d['decode']['ms_per_object'],3), 'dominant', r['kernel'][:48], round(r['achieved'],1), 'TF/s', 'all gemm', round(r['all_gemm_tflops'],1), 'step', round(d['ms_per_step'],2))"
  done
done | tee gpurun_out/r03_e_tile512_ab.txt
