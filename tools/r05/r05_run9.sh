#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=r05_i
timeout 900 python -m pytest tests/test_f16x3_gpu.py tests/test_epilogue_outputs_gpu.py tests/test_fused_splitk_gpu.py -q -x > gpurun_out/${T}_new_tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/${T}_new_tests.log
tail -5 gpurun_out/${T}_new_tests.log | cut -c1-200
tools/ab_bench.sh ${T}_up2 CS_NO_UP2_BATCH 1 4 7 14
tools/ab_bench.sh ${T}_gnfold CS_NO_GN_FOLD 1 7
tools/ab_bench.sh ${T}_static CS_NO_STATIC_SCALES 1 32
