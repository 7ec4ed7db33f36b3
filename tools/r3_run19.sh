cd $GRAFT_REPO_ROOT
for flag in "" "-DCS_PW_RING4"; do
  CS_EXTRA_HIPCC_FLAGS="$flag" python -m commonscenes_amd.build --force > /dev/null 2>&1
  for rep in 1 2; do
    timeout 300 python bench.py --no-cpu-baseline --no-fp32-leg --no-extras --steps 10 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ring[$flag]', 'ms/step', round(d['ms_per_step'],2))"
  done
  python tools/gemm_1tap.py 2>&1 | grep -v amdgpu | cut -c1-60
done | tee gpurun_out/r03_q_pw_ring4.txt
python -m commonscenes_amd.build --force > /dev/null 2>&1
