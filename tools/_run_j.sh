cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -k "unet or model or depth or epilogue or native" 2>&1 | tail -5
for rep in 1 2; do
for mode in on off; do
  if [ $mode = off ]; then export CS_NO_TOK_RULES=1; else unset CS_NO_TOK_RULES; fi
  for ob in 32 7 1; do
  timeout 600 python bench.py --no-cpu-baseline --no-traffic --no-extras --objects $ob --steps 12 --warmup 3 > gpurun_out/r04_j_bench_${mode}_$ob.json 2> /dev/null
  python -c "
import json; d=json.load(open('gpurun_out/r04_j_bench_${mode}_$ob.json')); print('$mode objects $ob: ms/step', round(d['ms_per_step'],3))"
  done
done
done
