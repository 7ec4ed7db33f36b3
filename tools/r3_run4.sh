cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for nb in 14 64; do
  echo "== SM_BATCH=$nb"
  SM_BATCH=$nb timeout 600 python tools/gemm_smallm.py 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r03_d_smallm_sweep.txt
timeout 300 python -m pytest tests/test_model_gpu.py -m gpu -q -k channel_split 2>&1 | tail -3
