#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -q -x --durations=5 > gpurun_out/r05_u_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05_u_tests.log )
tail -15 gpurun_out/r05_u_tests.log
bash tools/ab_bench.sh r05_u_wino CS_NO_WINO 1 7 32
