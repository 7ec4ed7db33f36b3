# round-3: split-K plan with any slice count (one round of 512 slots) vs the power-of-two rule, ms/step by batch, one box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for rep in 1 2; do for nobj in 1 2 4 7 14; do for arm in 1 ""; do
  CS_PLAN_POW2=$arm timeout 300 python bench.py --no-cpu-baseline --no-fp32-leg --no-extras --objects $nobj --steps 12 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('objects $nobj pow2=[$arm]', 'ms/step', round(d['ms_per_step'],3))"
done; done; done | tee gpurun_out/r03_aj_plan_any_ab.txt
( timeout 1200 python -m pytest tests/test_f16x3_gpu.py tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_unet_native_gpu.py tests/test_vqvae_native_gpu.py tests/test_c_host_gpu.py tests/test_parity_depth_gpu.py -m gpu -q -x > gpurun_out/r03_aj_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_aj_tests.log )
tail -4 gpurun_out/r03_aj_tests.log
