#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=r05_o
timeout 1200 python -m pytest tests/test_f16x3_gpu.py tests/test_parity_depth_gpu.py tests/test_model_gpu.py tests/test_epilogue_outputs_gpu.py -x -q 2>&1 | tail -4
for o in 7 32; do
for rep in 1 2; do
  for v in 0 1; do
    if [ $v = 1 ]; then export CS_NO_TOK_RULES=1; else unset CS_NO_TOK_RULES; fi
    st=20; [ "$o" -le 7 ] && st=40
    ms=$(timeout 600 python bench.py --no-cpu-baseline --no-traffic --no-extras --no-fp32-leg --no-gemm-profile --objects $o --steps $st --warmup 5 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "objects=$o CS_NO_TOK_RULES=$v rep=$rep ms_per_step=$ms" | tee -a gpurun_out/${T}_tok_rules_ab.txt
  done
done
done
