#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=r06i
( timeout 1800 python -m pytest tests -m gpu -q -x > gpurun_out/${TAG}_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_tests.log )
tail -4 gpurun_out/${TAG}_tests.log
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"
python - <<PY
import json
d = json.loads(open('gpurun_out/${TAG}_bench.json').read().strip().splitlines()[-1]); r = d['roofline']
print('steps/s', d['value'], 'ms', d['ms_per_step'], 'frac', r['frac'], 'avg_ms', r['avg_launch_ms'], 'launches', r['launches'], 'traffic ratio', r.get('traffic_over_algorithmic'), 'busy', r.get('mfma_busy_frac'))
print('c2', d['c2']['ms_per_step'], 'c7', d['c7']['ms_per_object_step'], 'decode/obj', d['decode']['ms_per_object'])
PY
