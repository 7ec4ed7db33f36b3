#!/bin/bash
# Same-box A/B of a what-if BUILD (tools/build_alt.sh) against the product library on the step loop:
#   tools/ab_lib.sh <tag> <alt library path (relative to the repo root)> [objects ...]
TAG=$1; ALT=$2; shift 2
OBJS=${@:-"32"}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/${TAG}_ab.txt
echo "# A/B of CS_LIB_PATH=$ALT (unset = product library) -- ms/step, same box, interleaved" > $OUT
for o in $OBJS; do
  st=20; [ "$o" -le 7 ] && st=40
  for rep in 1 2; do
    for v in 0 1; do
      if [ $v = 1 ]; then export CS_LIB_PATH=$PWD/$ALT; else unset CS_LIB_PATH; fi
      ms=$(timeout 600 python bench.py --no-cpu-baseline --no-traffic --no-extras --no-fp32-leg --no-gemm-profile --objects $o --steps $st --warmup 5 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
      echo "objects=$o alt=$v rep=$rep ms_per_step=$ms" | tee -a $OUT
    done
  done
done
unset CS_LIB_PATH
