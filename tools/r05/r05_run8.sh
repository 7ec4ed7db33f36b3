#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=r05_h
for rep in 1 2 3; do
  for v in 32 48; do
    ms=$(CS_SPLITK_CAP=$v timeout 600 python bench.py --no-cpu-baseline --no-traffic --no-extras --no-fp32-leg --no-gemm-profile --objects 1 --steps 60 --warmup 5 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "objects=1 CS_SPLITK_CAP=$v rep=$rep ms_per_step=$ms" | tee -a gpurun_out/${T}_splitk_cap_ab.txt
  done
done
tools/ab_bench.sh ${T}_fused CS_NO_FUSED_REDUCE 1
tools/ab_bench.sh ${T}_kwave CS_NO_KWAVE 1 7
