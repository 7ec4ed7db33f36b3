#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=r06b
( timeout 1500 python -m pytest tests -m gpu -q -x --durations=10 > gpurun_out/${TAG}_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_tests.log )
tail -22 gpurun_out/${TAG}_tests.log
bash tools/ab_lib.sh ${TAG}_fastsilu commonscenes_amd/alt/libcommonscenes_hip_fastsilu.so 32
timeout 900 python bench.py --no-cpu-baseline --no-traffic --no-extras --gemm-table > gpurun_out/${TAG}_bench_nocpu.json 2> gpurun_out/${TAG}_gemm_table.txt; echo "bench rc=$?"
grep -v "^[WEI]2026" gpurun_out/${TAG}_gemm_table.txt | head -48 > gpurun_out/${TAG}_gemm_table_clean.txt
head -22 gpurun_out/${TAG}_gemm_table_clean.txt
python - <<PY
import json
d = json.loads(open('gpurun_out/${TAG}_bench_nocpu.json').read().strip().splitlines()[-1]); r = d['roofline']
print('steps/s', d['value'], 'ms', d['ms_per_step'], 'frac', r['frac'], 'avg_ms', r['avg_launch_ms'], 'hooks', r.get('hooks_ms_per_step'), 'whole_frac', r.get('whole_step_executed_frac'))
PY
python tools/hbm_bench.py > gpurun_out/${TAG}_hbm_bound_kernels.txt 2>&1; tail -30 gpurun_out/${TAG}_hbm_bound_kernels.txt
CS_LIB_PATH=$PWD/commonscenes_amd/alt/libcommonscenes_hip_fastsilu.so python tools/hbm_bench.py > gpurun_out/${TAG}_hbm_bound_kernels_fastsilu.txt 2>&1; tail -30 gpurun_out/${TAG}_hbm_bound_kernels_fastsilu.txt
