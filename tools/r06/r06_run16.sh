#!/bin/bash
# A/B of the r6 one-tap tile rule (iv): CS_TOK_T2_MAXK=0 (rules ii / iii only) vs the default bound of 84 chunks
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/r06p_tok_t2_rule_ab.txt
echo "# A/B of one-tap rule (iv): CS_TOK_T2_MAXK=0 (= before) vs default -- ms/step, same box, interleaved" > $OUT
for o in 32 7 1; do
  st=20; [ "$o" -le 7 ] && st=40
  for rep in 1 2 3; do
    for v in 0 84; do
      ms=$(CS_TOK_T2_MAXK=$v timeout 600 python bench.py --no-cpu-baseline --no-traffic --no-extras --no-fp32-leg --no-gemm-profile --objects $o --steps $st --warmup 5 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
      echo "objects=$o CS_TOK_T2_MAXK=$v rep=$rep ms_per_step=$ms" | tee -a $OUT
    done
  done
done
