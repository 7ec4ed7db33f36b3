# round-3: folded Upsample classes store straight into the doubled grid -- tests, decode + bench A/B (CS_NO_UP2_DIRECT=1 = before)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1200 python -m pytest tests/test_f16x3_gpu.py tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_unet_native_gpu.py tests/test_vqvae_native_gpu.py tests/test_c_host_gpu.py tests/test_parity_depth_gpu.py -m gpu -q -x > gpurun_out/r03_y_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_y_tests.log )
tail -6 gpurun_out/r03_y_tests.log
for arm in "CS_NO_UP2_DIRECT=1" "CS_NO_UP2_DIRECT=" "CS_NO_UP2_DIRECT=1" "CS_NO_UP2_DIRECT="; do
  echo "== $arm"
  env $arm timeout 300 python tools/decode_bench.py 2>&1 | grep -v amdgpu.ids | head -1
  env $arm timeout 300 python bench.py --no-cpu-baseline --no-fp32-leg --steps 10 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench ms/step', round(d['ms_per_step'],2), 'c2', round(d['c2']['ms_per_step'],2), 'c7', round(d['c7']['ms_per_step'],2), 'decode', round(d['decode']['ms'],2))"
done | tee gpurun_out/r03_y_bench_ab.txt
