# round-3 GPU run 10: pre-split operands for the 128-row slab tile (medium / small batches): 1, 2, 4, 7, 16 objects
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for nobj in 7 1 2 4 16; do
  for arm in 0 1; do
    CS_SPLIT16_MIN_ROWS=$arm timeout 300 python bench.py --no-cpu-baseline --no-fp32-leg --no-extras --objects $nobj --steps 12 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('objects $nobj split16_small[$arm]', 'ms/step', round(d['ms_per_step'],3))"
  done
done | tee gpurun_out/r03_j_split16_small_ab.txt
timeout 600 python -m pytest tests/test_model_gpu.py tests/test_f16x3_gpu.py -m gpu -q 2>&1 | tail -3
