# round-3 GPU run 17: four-tap slab path for the folded Upsample convs: tests, then A/B on one box (step + decode)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1200 python -m pytest tests/test_f16x3_gpu.py tests/test_parity_depth_gpu.py tests/test_model_gpu.py tests/test_vqvae_native_gpu.py tests/test_unet_native_gpu.py -m gpu -q -x > gpurun_out/r03_o_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_o_tests.log )
tail -8 gpurun_out/r03_o_tests.log
for arm in 1 ""; do
  for rep in 1 2; do
    CS_NO_SLAB4=$arm timeout 300 python bench.py --no-cpu-baseline --no-fp32-leg --steps 10 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('noslab4[$arm]', 'ms/step', round(d['ms_per_step'],2), 'decode ms', round(d['decode']['ms'],2), 'c7', round(d['c7']['ms_per_step'],2))"
  done
done | tee gpurun_out/r03_o_slab4_ab.txt
CS_NO_SLAB4=1 timeout 300 python bench.py --no-cpu-baseline --no-fp32-leg --no-extras --gemm-table 2>&1 >/dev/null | grep "^ *12 " | head -4
timeout 300 python bench.py --no-cpu-baseline --no-fp32-leg --no-extras --gemm-table 2>&1 >/dev/null | grep "^ *12 " | head -4
timeout 300 python tools/decode_bench.py 2>&1 | grep -v amdgpu | head -8
