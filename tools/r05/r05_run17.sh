#!/bin/bash
# Winograd-W route: same-box A/B on the step loop (CS_NO_WINO=1 = the direct form everywhere) + the parity tests that run the UNet
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/ab_bench.sh r05_s_wino CS_NO_WINO 7 32
timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_parity_depth_gpu.py tests/test_full_size_fp64_gpu.py -x -q 2>&1 | tail -6
