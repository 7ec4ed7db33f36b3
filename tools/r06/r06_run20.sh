#!/bin/bash
# in-step A/B of two planner choices on a what-if build (-DCS_WHATIF_KNOBS): the fused gate on the 256-row tile; rule (i) off
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/r06x_planner_knobs_ab.txt
echo "# what-if build with planner knobs: base / CS_GEGLU_T4=1 (fused gate on the 256-row tile) / CS_NO_TOK_RULE1=1 -- ms/step at 32 objects, same box, interleaved" > $OUT
export CS_LIB_PATH=$PWD/commonscenes_amd/alt/libcommonscenes_hip_knobs.so
for rep in 1 2 3; do
  for v in base CS_GEGLU_T4 CS_NO_TOK_RULE1; do
    if [ $v = base ]; then ms=$(timeout 600 python bench.py --no-cpu-baseline --no-traffic --no-extras --no-fp32-leg --no-gemm-profile --objects 32 --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    else ms=$(env $v=1 timeout 600 python bench.py --no-cpu-baseline --no-traffic --no-extras --no-fp32-leg --no-gemm-profile --objects 32 --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"); fi
    echo "$v rep=$rep ms_per_step=$ms" | tee -a $OUT
  done
done
