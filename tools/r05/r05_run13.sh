#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=r05_m
REPO=$(pwd)
python tools/tok_phase.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${T}_tok_phase.txt
cd /tmp && export TMPDIR=/tmp
for v in product ablate1024 ablate2048 ablate3072; do
  if [ $v = product ]; then unset CS_LIB; else export CS_LIB=$REPO/variants/libcs_$v.so; fi
  timeout 300 rocprofv3 --kernel-trace -d $REPO/gpurun_out/prof_${T}_$v -o t -- python $REPO/tools/tok_phase.py > /dev/null 2>&1
  DB=$(find $REPO/gpurun_out/prof_${T}_$v -name "*.db" | head -1)
  echo "== $v" | tee -a $REPO/gpurun_out/${T}_tok_phase.txt
  python $REPO/tools/rocpd_sequence.py $DB conv_gemm_f16x3 20 2>&1 | tee -a $REPO/gpurun_out/${T}_tok_phase.txt
  rm -rf $REPO/gpurun_out/prof_${T}_$v
done
