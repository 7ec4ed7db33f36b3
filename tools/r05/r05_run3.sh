#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=r05_c
timeout 900 python -m pytest tests/test_kwave_gpu.py -x -q -s > gpurun_out/${T}_kw_tests.log 2>&1; echo "kw tests rc=$?" | tee -a gpurun_out/${T}_kw_tests.log
grep -v "^$" gpurun_out/${T}_kw_tests.log | tail -40 | cut -c1-220
timeout 1800 python -m pytest tests -m gpu -q -x > gpurun_out/${T}_tests.log 2>&1; echo "suite rc=$?" | tee -a gpurun_out/${T}_tests.log
tail -12 gpurun_out/${T}_tests.log | cut -c1-220
tools/ab_bench.sh ${T}_kwave CS_NO_KWAVE 1 7
timeout 900 python tools/eval_walkthrough.py --scenes 8 --samples 1 --width 224 --ddim-steps 20 --points 2000 --batch-scenes > gpurun_out/${T}_walkthrough_batch_scenes.txt 2>&1
grep EVAL_WALKTHROUGH gpurun_out/${T}_walkthrough_batch_scenes.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read().split('EVAL_WALKTHROUGH ',1)[1]); print(json.dumps(d.get('batch_scenes'), indent=1))"
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_${T}3 -o bench -- python $REPO/bench.py --no-cpu-baseline --no-traffic --no-extras --no-fp32-leg --objects 1 --steps 20 --warmup 3 > $REPO/gpurun_out/${T}_bench_c2_under_rocprof.json 2> /dev/null
cd $REPO
DB=$(find gpurun_out/prof_${T}3 -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB > gpurun_out/${T}_c2_kernel_stats.txt
rm -rf gpurun_out/prof_${T}3
head -24 gpurun_out/${T}_c2_kernel_stats.txt | cut -c1-200
