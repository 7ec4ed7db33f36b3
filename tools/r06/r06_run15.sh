#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=r06o
( timeout 900 python -m pytest tests/test_f16x3_gpu.py tests/test_ops_gpu.py -m gpu -q -x -k "attention" > gpurun_out/${TAG}_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_tests.log )
tail -3 gpurun_out/${TAG}_tests.log
python tools/attn_bench.py 2>&1 | grep -v amdgpu | cut -c1-110 | tee gpurun_out/${TAG}_attn_bench.txt
bash tools/attn_pmc.sh gpurun_out/${TAG}_attn_pmc | grep -E "INSTS_VALU|WAVE_CYCLES|GRBM|ACTIVE_INST_VALU"
