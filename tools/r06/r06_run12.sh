#!/bin/bash
# what-if: Winograd position GEMMs on two 128-row workgroups per CU (-DCS_WINO_T2 build + CS_WINO_T2=1) vs the product's 256-row tile
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export CS_WINO_T2=1
bash tools/ab_lib.sh r06l_wino_tile2_whatif commonscenes_amd/alt/libcommonscenes_hip_t2.so 32 7 1
( CS_LIB_PATH=$PWD/commonscenes_amd/alt/libcommonscenes_hip_t2.so timeout 900 python -m pytest tests/test_wino_gpu.py -m gpu -q -x > gpurun_out/r06l_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r06l_tests.log )
tail -4 gpurun_out/r06l_tests.log
