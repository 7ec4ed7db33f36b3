#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for o in 1 2; do for rep in 1 2 3; do for mode in default min512; do
  unset CS_WINO_MIN_ROWS; [ $mode = min512 ] && export CS_WINO_MIN_ROWS=512
  ms=$(timeout 600 python bench.py --no-cpu-baseline --no-traffic --no-extras --no-fp32-leg --no-gemm-profile --objects $o --steps 40 --warmup 5 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
  echo "objects=$o wino_min_rows=$mode rep=$rep ms_per_step=$ms" | tee -a gpurun_out/r05_ag_wino_min512_ab.txt
done; done; done
