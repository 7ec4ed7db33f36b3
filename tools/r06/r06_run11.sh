#!/bin/bash
# restored-state check: full GPU suite (with the new C3 full-depth batch test) + the default bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=${1:-r06k}
( timeout 1700 python -m pytest tests -m gpu -q -x --durations=8 > gpurun_out/${TAG}_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_tests.log )
tail -14 gpurun_out/${TAG}_tests.log
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/${TAG}_bench_nocpu.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"
python - <<PY
import json
d = json.load(open('gpurun_out/${TAG}_bench_nocpu.json')); r = d['roofline']
print('steps/s', d['value'], 'ms', d['ms_per_step'], 'frac', r['frac'], 'traffic', r['traffic'])
print('c2', d['c2']['ms_per_step'], 'c7', d['c7']['ms_per_step'], 'decode/obj', d['decode']['ms_per_object'])
PY
