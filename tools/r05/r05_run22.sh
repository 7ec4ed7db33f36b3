#!/bin/bash
# F(4,3) along W: tests, then same-box A/B against F(2,3) (CS_NO_WINO43=1) on the step loop and per conv
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_wino_gpu.py -x -q 2>&1 | tail -5
for v in 0 1; do
  [ $v = 1 ] && export CS_NO_WINO43=1 || unset CS_NO_WINO43
  echo "== CS_NO_WINO43=$v" | tee -a gpurun_out/r05_aa_wino43_bench.txt
  python tools/wino_bench.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r05_aa_wino43_bench.txt
done
unset CS_NO_WINO43
bash tools/ab_bench.sh r05_aa_wino43 CS_NO_WINO43 1 7 32
