cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/tok_whatif3.sh 2>&1 | tee gpurun_out/r03_h_tok_whatif3.txt
( timeout 900 python -m pytest tests/test_parity_depth_gpu.py tests/test_unet_native_gpu.py tests/test_vqvae_native_gpu.py -m gpu -q > gpurun_out/r03_h_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_h_tests.log )
tail -5 gpurun_out/r03_h_tests.log
