#!/bin/bash
# what-if: the slab kernel with every second / only the first B fragment pair read from LDS (timing only)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=r05_n
for rep in 1 2; do
for v in product ablate4096 ablate8192; do
  if [ $v = product ]; then unset CS_LIB; else export CS_LIB=$PWD/variants/libcs_$v.so; fi
  echo "== $v" | tee -a gpurun_out/${T}_lds_whatif.txt
  python tools/conv_pre_bench.py --pre-only 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/${T}_lds_whatif.txt
done
done
