#!/bin/bash
# phase budget of the Winograd position GEMMs at batch 64: product / no epilogue / no K loop (timing-only builds), GPU-side durations
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
for v in ${VARIANTS:-product ablate1024 ablate2048}; do
  if [ $v = product ]; then unset CS_LIB; else export CS_LIB=$REPO/variants/libcs_$v.so; fi
  timeout 300 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_ph -o t -- python $REPO/tools/wino_bench.py > /dev/null 2>&1
  DB=$(find $REPO/gpurun_out/prof_ph -name "*.db" | head -1)
  echo "== $v" | tee -a $REPO/gpurun_out/r05_af_wino_phase.txt
  python $REPO/tools/rocpd_by_grid.py $DB "conv_gemm_f16x3_kernel<1, 7, 8, 1, true, 32, false, 3," 2>&1 | tee -a $REPO/gpurun_out/r05_af_wino_phase.txt
  rm -rf $REPO/gpurun_out/prof_ph
done
