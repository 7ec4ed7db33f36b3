#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=r05_d
timeout 300 python tools/kw_debug.py > gpurun_out/${T}_kw_debug.txt 2>&1; tail -20 gpurun_out/${T}_kw_debug.txt
timeout 900 python -m pytest tests/test_kwave_gpu.py "tests/test_parity_depth_gpu.py::test_transformer_operands_take_static_scales_and_need_no_fallback" tests/test_parity_depth_gpu.py::test_rel2shape_reruns_an_overflowing_minibatch_in_fp32 -q -s > gpurun_out/${T}_new_tests.log 2>&1; echo "new tests rc=$?" | tee -a gpurun_out/${T}_new_tests.log
grep -v "^$" gpurun_out/${T}_new_tests.log | grep -v Warning | tail -60 | cut -c1-260
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/${T}_tests.log 2>&1; echo "suite rc=$?" | tee -a gpurun_out/${T}_tests.log
tail -15 gpurun_out/${T}_tests.log | cut -c1-220
timeout 600 python tools/check_checkpoint.py --synthetic --steps 6 > gpurun_out/${T}_check_checkpoint_synthetic.txt 2>&1; echo "check rc=$?"
grep -v CHECK_CHECKPOINT gpurun_out/${T}_check_checkpoint_synthetic.txt | tail -30 | cut -c1-240
tools/ab_bench.sh ${T}_static CS_NO_STATIC_SCALES 1 32
timeout 900 python tools/eval_walkthrough.py --scenes 6 --samples 1 --width 224 --ddim-steps 50 --points 2000 --batch-scenes > gpurun_out/${T}_walkthrough_batch_scenes.txt 2>&1
grep EVAL_WALKTHROUGH gpurun_out/${T}_walkthrough_batch_scenes.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read().split('EVAL_WALKTHROUGH ',1)[1]); print(json.dumps(d.get('batch_scenes')))"
