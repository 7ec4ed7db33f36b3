#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -q -x --durations=5 > gpurun_out/r05_x_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05_x_tests.log )
tail -14 gpurun_out/r05_x_tests.log
bash tools/ab_bench.sh r05_x_wino CS_NO_WINO 1 7 32
python tools/decode_bench.py 2>&1 | grep -v amdgpu.ids | head -20 | tee gpurun_out/r05_x_decode_table.txt
CS_NO_WINO=1 python tools/decode_bench.py 2>&1 | grep -v amdgpu.ids | head -3 | tee -a gpurun_out/r05_x_decode_table.txt
