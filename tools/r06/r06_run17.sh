#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/r06p_tok_t2_rule_sweep.txt
echo "# sweep of the one-tap rule (iv) bound CS_TOK_T2_MAXK (chunks of 16) -- ms/step at 32 objects, same box, interleaved" > $OUT
for rep in 1 2 3; do
  for v in 84 112 128 176 400; do
    ms=$(CS_TOK_T2_MAXK=$v timeout 600 python bench.py --no-cpu-baseline --no-traffic --no-extras --no-fp32-leg --no-gemm-profile --objects 32 --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "CS_TOK_T2_MAXK=$v rep=$rep ms_per_step=$ms" | tee -a $OUT
  done
done
