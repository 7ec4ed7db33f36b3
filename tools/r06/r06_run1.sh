#!/bin/bash
# r6 GPU call 1: suite, same-box A/B of the Winograd tail plan (CS_NO_WINO_TAIL) at 32 objects, default bench line, GEMM table
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=r06a
( timeout 1500 python -m pytest tests -m gpu -q -x --durations=10 > gpurun_out/${TAG}_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_tests.log )
tail -25 gpurun_out/${TAG}_tests.log
bash tools/ab_bench.sh ${TAG}_wino_tail CS_NO_WINO_TAIL 32
timeout 900 python bench.py --no-cpu-baseline --no-traffic --gemm-table > gpurun_out/${TAG}_bench_nocpu.json 2> gpurun_out/${TAG}_gemm_table.txt; echo "bench rc=$?"
grep -v "^[WEI]2026" gpurun_out/${TAG}_gemm_table.txt | head -48 > gpurun_out/${TAG}_gemm_table_clean.txt
cat gpurun_out/${TAG}_gemm_table_clean.txt | head -30
python - <<PY
import json
d = json.loads(open('gpurun_out/${TAG}_bench_nocpu.json').read().strip().splitlines()[-1]); r = d['roofline']
print('steps/s', d['value'], 'ms', d['ms_per_step'], 'frac', r['frac'], 'avg_ms', r['avg_launch_ms'], 'hooks', r.get('hooks_ms_per_step'), 'whole_frac', r.get('whole_step_executed_frac'))
print('c2', d['c2']['ms_per_step'], 'c7', d['c7']['ms_per_step'], d['c7']['ms_per_object_step'], 'decode/obj', d['decode']['ms_per_object'])
PY
