# round-3 GPU run 2: -m gpu suite with the channel-split ResBlocks (both drivers), then the A/B on one box:
# CS_NO_CFG_SPLIT=1 (r2 form: the whole concatenation convolved at the CFG batch) vs the default
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( CS_PARITY_LOG=$PWD/gpurun_out/r03_b_parity_log.txt timeout 1500 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/r03_b_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_b_tests.log )
tail -15 gpurun_out/r03_b_tests.log
for arm in 1 ""; do
  for rep in 1 2; do
    CS_NO_CFG_SPLIT=$arm timeout 300 python bench.py --no-cpu-baseline --no-fp32-leg --steps 10 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nosplit[$arm]', 'ms/step', round(d['ms_per_step'],2), 'frac', round(d['roofline']['frac'],4), 'exec TF/s', round(d['roofline']['whole_step_executed_tflops'],1), 'c2', round(d['c2']['ms_per_step'],2), 'c7', round(d['c7']['ms_per_step'],2), 'e2e', round(d['end_to_end']['value'],3))"
  done
done | tee gpurun_out/r03_b_cfgsplit_ab.txt
CS_NO_CFG_SPLIT= timeout 300 python bench.py --no-cpu-baseline --no-fp32-leg --driver native --steps 10 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('native driver ms/step', round(d['ms_per_step'],2))" | tee -a gpurun_out/r03_b_cfgsplit_ab.txt
