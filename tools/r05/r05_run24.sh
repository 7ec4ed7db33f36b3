#!/bin/bash
# F(4,3) threshold: default 4096 rows against 1024 (and the F(2,3) threshold at 512 with it), small batches, same box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=r05_ac
for o in 1 2 4 7; do
for rep in 1 2; do
  for mode in default w43_1024 w43_2048; do
    unset CS_WINO43_MIN_ROWS
    [ $mode = w43_1024 ] && export CS_WINO43_MIN_ROWS=1024
    [ $mode = w43_2048 ] && export CS_WINO43_MIN_ROWS=2048
    ms=$(timeout 600 python bench.py --no-cpu-baseline --no-traffic --no-extras --no-fp32-leg --no-gemm-profile --objects $o --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "objects=$o wino43_min_rows=$mode rep=$rep ms_per_step=$ms" | tee -a gpurun_out/${T}_wino43_minrows_ab.txt
  done
done
done
