#!/bin/bash
# Winograd-W route at small batches: direct form / default threshold / threshold 1024 rows, same box, interleaved
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=r05_t
for o in 1 2 4 7 14; do
for rep in 1 2; do
  for mode in direct default min1024; do
    unset CS_NO_WINO CS_WINO_MIN_ROWS
    [ $mode = direct ] && export CS_NO_WINO=1
    [ $mode = min1024 ] && export CS_WINO_MIN_ROWS=1024
    ms=$(timeout 600 python bench.py --no-cpu-baseline --no-traffic --no-extras --no-fp32-leg --no-gemm-profile --objects $o --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "objects=$o wino=$mode rep=$rep ms_per_step=$ms" | tee -a gpurun_out/${T}_wino_small_ab.txt
  done
done
done
