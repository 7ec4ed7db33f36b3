#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( CS_PARITY_LOG=$PWD/gpurun_out/r05_ab_parity_log.txt timeout 1500 python -m pytest tests -m gpu -q -x --durations=5 > gpurun_out/r05_ab_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05_ab_tests.log )
tail -14 gpurun_out/r05_ab_tests.log
python tools/wino_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_ab_wino43_bench.txt
