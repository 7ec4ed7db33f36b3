#!/bin/bash
# rule (iv) at the mid-size batches real scenes produce (4 / 14 / 21 objects): CS_TOK_T2_MAXK=0 (= before) vs default
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/r06w_tok_t2_rule_midsize_ab.txt
echo "# one-tap rule (iv) at mid-size batches: CS_TOK_T2_MAXK=0 (= before) vs default (128) -- ms/step, same box, interleaved" > $OUT
for o in 4 14 21; do
  for rep in 1 2; do
    for v in 0 128; do
      ms=$(CS_TOK_T2_MAXK=$v timeout 600 python bench.py --no-cpu-baseline --no-traffic --no-extras --no-fp32-leg --no-gemm-profile --objects $o --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
      echo "objects=$o CS_TOK_T2_MAXK=$v rep=$rep ms_per_step=$ms" | tee -a $OUT
    done
  done
done
