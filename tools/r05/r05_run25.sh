#!/bin/bash
# K slices of the position GEMMs: the cost model's choice against forced counts, per conv shape at batch 64 and 14
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for b in 64 14; do
for sp in 0 1 2 3 4 6; do
  [ $sp = 0 ] && unset CS_WINO_SPLITS || export CS_WINO_SPLITS=$sp
  echo "== batch $b CS_WINO_SPLITS=$sp" | tee -a gpurun_out/r05_ae_wino_splits.txt
  python tools/wino_bench.py $b 2>&1 | grep -v amdgpu.ids | sed 's/| GroupNorm.*//' | tee -a gpurun_out/r05_ae_wino_splits.txt
done
done
