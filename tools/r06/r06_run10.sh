#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=r06j
bash tools/ab_lib.sh ${TAG}_gn_wino_v4 commonscenes_amd/alt/libcommonscenes_hip_gnv4.so 32
( timeout 1500 python -m pytest tests/test_wino_gpu.py tests/test_model_gpu.py tests/test_unet_native_gpu.py tests/test_full_size_fp64_gpu.py -m gpu -q -x > gpurun_out/${TAG}_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_tests.log )
tail -4 gpurun_out/${TAG}_tests.log
REPO=$(pwd); cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_j -o bench -- python $REPO/bench.py --no-cpu-baseline --no-extras --no-fp32-leg --steps 10 --warmup 3 > /dev/null 2>&1
cd $REPO
DB=$(find /tmp/prof_j -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB | head -14 | cut -c1-150
