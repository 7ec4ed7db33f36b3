#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=r05_g
timeout 900 python -m pytest tests/test_sharded_gpu.py tests/test_kwave_gpu.py -q > gpurun_out/${T}_new_tests.log 2>&1; echo "new tests rc=$?" | tee -a gpurun_out/${T}_new_tests.log
tail -5 gpurun_out/${T}_new_tests.log | cut -c1-200
timeout 900 python tools/graph_ab.py 1 2 4 7 > gpurun_out/${T}_graph_ab.txt 2>&1; tail -6 gpurun_out/${T}_graph_ab.txt
for drv in python native; do
  for o in 1 7; do
    ms=$(timeout 600 python bench.py --no-cpu-baseline --no-traffic --no-extras --no-fp32-leg --driver $drv --objects $o --steps 40 --warmup 5 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "driver=$drv objects=$o ms_per_step=$ms" | tee -a gpurun_out/${T}_driver_ab.txt
  done
done
