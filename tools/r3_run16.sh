cd $GRAFT_REPO_ROOT
for nobj in 7 4 14; do
  for mr in 65536 32768 16384; do
    CS_CFG_SPLIT_MIN_ROWS=$mr timeout 300 python bench.py --no-cpu-baseline --no-fp32-leg --no-extras --objects $nobj --steps 12 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('objects $nobj cfg-split min rows $mr', 'ms/step', round(d['ms_per_step'],3))"
  done
done | tee gpurun_out/r03_n_cfgsplit_threshold.txt
