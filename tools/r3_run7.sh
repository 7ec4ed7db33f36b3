# round-3 GPU run 7: suite on the producer-derived operand scales; per-shape decode tables with / without the 512-row tiles
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( CS_PARITY_LOG=$PWD/gpurun_out/r03_g_parity_log.txt timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r03_g_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_g_tests.log )
tail -8 gpurun_out/r03_g_tests.log
grep -h "tiny-gamma\|heavy-tailed\|uniformly" gpurun_out/r03_g_tests.log | head
for arm in 1 9 ""; do
  echo "== CS_NO_TILE512=$arm"
  CS_NO_TILE512=$arm timeout 300 python tools/decode_bench.py 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r03_g_decode_tables.txt
timeout 300 python bench.py --no-cpu-baseline --no-fp32-leg --steps 10 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench ms/step', round(d['ms_per_step'],2), 'c2', round(d['c2']['ms_per_step'],2), 'c7', round(d['c7']['ms_per_step'],2), 'decode', round(d['decode']['ms'],2))"
