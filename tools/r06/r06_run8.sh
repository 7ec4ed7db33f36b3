#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=r06h
ALT=commonscenes_amd/alt/libcommonscenes_hip_pipedplain.so
bash tools/ab_lib.sh ${TAG}_piped_plain $ALT 32 7 1
( CS_LIB_PATH=$PWD/$ALT timeout 1500 python -m pytest tests/test_f16x3_gpu.py tests/test_wino_gpu.py tests/test_model_gpu.py tests/test_ops_gpu.py tests/test_unet_native_gpu.py -m gpu -q -x > gpurun_out/${TAG}_tests_alt.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_tests_alt.log )
tail -5 gpurun_out/${TAG}_tests_alt.log
CS_LIB_PATH=$PWD/$ALT timeout 600 python bench.py --no-cpu-baseline --no-extras --no-fp32-leg --steps 10 --warmup 3 --gemm-table > gpurun_out/${TAG}_bench_alt.json 2> gpurun_out/${TAG}_gemm_table_alt.txt
grep -v "^[WEI]2026" gpurun_out/${TAG}_gemm_table_alt.txt | head -30
