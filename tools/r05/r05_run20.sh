#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=r05_v
timeout 600 python -m pytest tests/test_wino_gpu.py -x -q 2>&1 | tail -3
timeout 900 python bench.py --no-cpu-baseline --no-traffic --no-extras --no-fp32-leg --gemm-table > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_gemm_table.txt
grep -v "^[WEI]2026\|amdgpu.ids" gpurun_out/${T}_gemm_table.txt | head -48
python - <<PY
import json
d = json.loads(open('gpurun_out/${T}_bench.json').read().strip().splitlines()[-1]); r = d['roofline']
print('steps/s', d['value'], 'ms', d['ms_per_step'])
for k in ('kernel','rocprof_kernel','achieved','frac','launches','avg_launch_ms','algorithmic_gflop_per_launch','algorithmic_bytes_per_launch','winograd_direct_equivalent_tflops','winograd_output_transform_ms_per_step_share','share_of_wall_time','all_gemm_tflops','all_gemm_direct_equivalent_tflops','all_gemm_share_of_wall_time'):
    print(' ', k, r.get(k))
PY
for mode in default min512; do for o in 1 2; do for rep in 1 2; do
  unset CS_WINO_MIN_ROWS; [ $mode = min512 ] && export CS_WINO_MIN_ROWS=512
  ms=$(timeout 600 python bench.py --no-cpu-baseline --no-traffic --no-extras --no-fp32-leg --no-gemm-profile --objects $o --steps 40 --warmup 5 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
  echo "objects=$o wino_min_rows=$mode rep=$rep ms_per_step=$ms" | tee -a gpurun_out/${T}_wino_min512_ab.txt
done; done; done
