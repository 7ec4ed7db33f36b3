#!/bin/bash
# K-sliced slab convs at one object and at the mini-batch of 7: 256-row tiles (product) vs two 128-row workgroups per CU
# (CS_SLICE_TILE2=1) -- whole kernel and K loop alone (no-epilogue build): what an in-workgroup K split would run at
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=r05_p
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
for nb in 2 14; do
for v in product ablate1024; do
  if [ $v = product ]; then unset CS_LIB; else export CS_LIB=$REPO/variants/libcs_$v.so; fi
  for t2 in 0 1; do
    [ $t2 = 1 ] && export CS_SLICE_TILE2=1 || unset CS_SLICE_TILE2
    CP_BATCH=$nb timeout 300 rocprofv3 --kernel-trace -d $REPO/gpurun_out/prof_${T} -o t -- python $REPO/tools/conv_phase.py > /dev/null 2>&1
    DB=$(find $REPO/gpurun_out/prof_${T} -name "*.db" | head -1)
    echo "== batch $nb $v slice_tile2=$t2" | tee -a $REPO/gpurun_out/${T}_slice_tile2.txt
    python $REPO/tools/rocpd_sequence.py $DB conv_gemm_f16x3 20 2>&1 | tee -a $REPO/gpurun_out/${T}_slice_tile2.txt
    rm -rf $REPO/gpurun_out/prof_${T}
  done
done
done
