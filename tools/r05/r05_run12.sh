#!/bin/bash
# phase budget of the K-sliced slab convs at one object: product vs no-epilogue / no-K-loop / neither builds
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=r05_l
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
for v in product ablate1024 ablate2048 ablate3072; do
  if [ $v = product ]; then unset CS_LIB; else export CS_LIB=$REPO/variants/libcs_$v.so; fi
  for fused in 0 1; do
    [ $fused = 0 ] && export CS_NO_FUSED_REDUCE=1 || unset CS_NO_FUSED_REDUCE
    timeout 300 rocprofv3 --kernel-trace -d $REPO/gpurun_out/prof_${T}_$v$fused -o t -- python $REPO/tools/conv_phase.py > /dev/null 2>&1
    DB=$(find $REPO/gpurun_out/prof_${T}_$v$fused -name "*.db" | head -1)
    echo "== $v fused_reduce=$fused" | tee -a $REPO/gpurun_out/${T}_conv_phase.txt
    python $REPO/tools/rocpd_sequence.py $DB conv_gemm_f16x3 20 2>&1 | tee -a $REPO/gpurun_out/${T}_conv_phase.txt
    [ $fused = 0 ] && python $REPO/tools/rocpd_sequence.py $DB splitk_reduce 20 2>&1 | tee -a $REPO/gpurun_out/${T}_conv_phase.txt
    rm -rf $REPO/gpurun_out/prof_${T}_$v$fused
  done
done
