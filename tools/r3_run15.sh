cd $GRAFT_REPO_ROOT
for nobj in 7 14 4 5 6 3; do
  for lim in 512 704; do
    CS_PLAN_LONGK_LIMIT=$lim timeout 300 python bench.py --no-cpu-baseline --no-fp32-leg --no-extras --objects $nobj --steps 12 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('objects $nobj longK limit $lim', 'ms/step', round(d['ms_per_step'],3))"
  done
done | tee gpurun_out/r03_m_plan_longk_ab.txt
