#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=r06d
( timeout 1800 python -m pytest tests/test_parity_depth_gpu.py tests/test_rccl_single_rank_gpu.py tests/test_sharded_gpu.py tests/test_unet_native_gpu.py tests/test_vae_facade_gpu.py tests/test_vqvae_native_gpu.py tests/test_wino_gpu.py tests/test_metrics_gpu.py -m gpu -q --durations=10 -s > gpurun_out/${TAG}_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_tests.log )
tail -30 gpurun_out/${TAG}_tests.log
grep -h "static scales \[\|VQ attention\|concat AttentionBlock\|conv_in operand\|tail plan\|traj100\|e2e100\|FAILED" gpurun_out/${TAG}_tests.log | head -40
