#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=r06e
( timeout 1800 python -m pytest tests -m gpu -q -x --durations=5 > gpurun_out/${TAG}_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_tests.log )
tail -12 gpurun_out/${TAG}_tests.log
REPO=$(pwd)
OUT=$REPO/gpurun_out/${TAG}_tok_phase.txt
: > $OUT
cd /tmp && export TMPDIR=/tmp
for v in product ablate1024 ablate2048 ablate18432 ablate34816; do
  if [ $v = product ]; then unset CS_LIB; else export CS_LIB=$REPO/commonscenes_amd/alt/libcommonscenes_hip_$v.so; fi
  rm -rf /tmp/tp_$v
  TP_TILES=2 TP_N=12 timeout 300 rocprofv3 --kernel-trace -d /tmp/tp_$v -o t -- python $REPO/tools/tok_phase.py > /tmp/tp_$v.log 2>&1
  DB=$(find /tmp/tp_$v -name "*.db" | head -1)
  echo "== $v" >> $OUT
  grep "^M=" /tmp/tp_$v.log >> $OUT
  [ -n "$DB" ] && python $REPO/tools/rocpd_sequence.py $DB conv_gemm_f16x3 12 >> $OUT
done
unset CS_LIB
cat $OUT
