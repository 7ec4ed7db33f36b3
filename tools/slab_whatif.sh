# What-if for an A-operand slab shared by the kw (and kh) taps: CS_ABLATE=128 drops the A DMA of taps with kw != 0,
# 256 of all taps but one per kd (results wrong, timing only).  Same box, conv shapes of tools/gemm_bench.py.
cd $GRAFT_REPO_ROOT
for ab in 0 128 256 1 0; do
  CS_EXTRA_HIPCC_FLAGS="-DCS_ABLATE=$ab" python -m commonscenes_amd.build --force > /dev/null 2>&1
  echo "== CS_ABLATE=$ab"
  for i in 0 1 3 5; do python tools/gemm_bench.py --math f16x3 --only $i --iters 10 2>&1 | grep -v amdgpu.ids; done
done
python -m commonscenes_amd.build --force > /dev/null 2>&1
