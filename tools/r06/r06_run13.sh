#!/bin/bash
# r6: the pipelined eight-wave attention kernel (attn_f16x3_pp_kernel) -- bit-equality test, kernel timings, step A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=r06m
( timeout 900 python -m pytest tests/test_f16x3_gpu.py -m gpu -q -x -k "attention" > gpurun_out/${TAG}_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_tests.log )
tail -5 gpurun_out/${TAG}_tests.log
{ echo "== product (pipelined)"; python tools/attn_bench.py; echo "== CS_NO_ATTN_PP=1"; CS_NO_ATTN_PP=1 python tools/attn_bench.py; } > gpurun_out/${TAG}_attn_bench.txt 2>&1
grep -v amdgpu.ids gpurun_out/${TAG}_attn_bench.txt | cut -c1-150
bash tools/ab_bench.sh ${TAG}_attn_pp CS_NO_ATTN_PP 32
