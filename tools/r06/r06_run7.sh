#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/ab_bench.sh r06g_c2_slice_tile2 CS_SLICE_TILE2 1 2
bash tools/ab_bench.sh r06g_c2_no_wino CS_NO_WINO 1 2
bash tools/ab_bench.sh r06g_c2_no_fused_reduce CS_NO_FUSED_REDUCE 1 2
timeout 600 python bench.py --no-cpu-baseline --no-extras --no-fp32-leg --steps 10 --warmup 3 > gpurun_out/r06g_bench_steploop.json 2>/dev/null
python - <<PY
import json
d=json.loads(open('gpurun_out/r06g_bench_steploop.json').read().strip().splitlines()[-1]); r=d['roofline']
print({k: r.get(k) for k in ('launches','host_calls','avg_launch_ms','algorithmic_bytes_per_launch','traffic','traffic_over_algorithmic','frac')}, d['ms_per_step'])
PY
