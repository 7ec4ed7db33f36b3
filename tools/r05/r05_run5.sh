#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=r05_e
timeout 900 python -m pytest tests/test_kwave_gpu.py "tests/test_parity_depth_gpu.py::test_transformer_operands_take_static_scales_and_need_no_fallback" tests/test_vae_facade_gpu.py tests/test_model_gpu.py::test_vq_decode_batch_invariance tests/test_vqvae_native_gpu.py -q -s > gpurun_out/${T}_new_tests.log 2>&1; echo "new tests rc=$?" | tee -a gpurun_out/${T}_new_tests.log
grep -v "^$" gpurun_out/${T}_new_tests.log | grep -v Warning | tail -40 | cut -c1-300
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/${T}_tests.log 2>&1; echo "suite rc=$?" | tee -a gpurun_out/${T}_tests.log
tail -15 gpurun_out/${T}_tests.log | cut -c1-220
for rep in 1 2; do
  for v in 256 1024; do
    ms=$(CS_KWAVE_MAX_TILES=$v timeout 600 python bench.py --no-cpu-baseline --no-traffic --no-extras --no-fp32-leg --objects 7 --steps 40 --warmup 5 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "objects=7 CS_KWAVE_MAX_TILES=$v rep=$rep ms_per_step=$ms" | tee -a gpurun_out/${T}_kwave_tiles_ab.txt
  done
done
for rep in 1 2; do
  for v in 0 1; do
    if [ $v = 1 ]; then export CS_NO_GN_PARTS=1; else unset CS_NO_GN_PARTS; fi
    timeout 600 python tools/decode_bench.py 2>/dev/null | head -3 | sed "s/^/CS_NO_GN_PARTS=$v rep=$rep: /" | tee -a gpurun_out/${T}_decode_gn_parts_ab.txt
  done
done
unset CS_NO_GN_PARTS
timeout 1200 python tools/eval_walkthrough.py --scenes 8 --samples 1 --width 224 --ddim-steps 100 --points 2000 --batch-scenes > gpurun_out/${T}_walkthrough_batch_scenes.txt 2>&1
grep EVAL_WALKTHROUGH gpurun_out/${T}_walkthrough_batch_scenes.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read().split('EVAL_WALKTHROUGH ',1)[1]); print(json.dumps(d.get('batch_scenes')))"
