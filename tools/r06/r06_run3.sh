#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=r06c
( timeout 1800 python -m pytest tests -m gpu -q -x --durations=10 > gpurun_out/${TAG}_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_tests.log )
tail -22 gpurun_out/${TAG}_tests.log
grep -h "static scales\|VQ attention\|concat AttentionBlock\|conv_in operand\|tail plan\|traj100\|e2e100" gpurun_out/${TAG}_tests.log | head -30
timeout 600 python tools/wino43x23_proto.py > gpurun_out/${TAG}_wino43x23_numerics.txt 2>&1; cat gpurun_out/${TAG}_wino43x23_numerics.txt | tail -8
bash tools/ab_lib.sh ${TAG}_ocml_erf commonscenes_amd/alt/libcommonscenes_hip_ocmlerf.so 32
