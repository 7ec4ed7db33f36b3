#!/bin/bash
# Same-box A/B of one CsDebug switch on the step loop: tools/ab_bench.sh <tag> <ENV_VAR> [objects ...]
#   runs bench.py --no-extras at each object count with the switch unset and set (=1), twice each, interleaved.
TAG=$1; VAR=$2; shift 2
OBJS=${@:-"1 7 32"}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/${TAG}_ab.txt
echo "# A/B of $VAR (unset = product path) -- ms/step, same box, interleaved" > $OUT
for o in $OBJS; do
  st=20; [ "$o" -le 7 ] && st=40
  for rep in 1 2; do
    for v in 0 1; do
      if [ $v = 1 ]; then export $VAR=1; else unset $VAR; fi
      ms=$(timeout 600 python bench.py --no-cpu-baseline --no-traffic --no-extras --no-fp32-leg --no-gemm-profile --objects $o --steps $st --warmup 5 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
      echo "objects=$o $VAR=$v rep=$rep ms_per_step=$ms" | tee -a $OUT
    done
  done
done
unset $VAR
