# What-if on the slab kernel: CS_ABLATE=512 feeds the raw LDS bits to the MFMAs (no fp32 -> hi/lo conversion VALU; results
# wrong, timing only) -- the upper bound of what producer-side pre-split activations could give it.
cd $GRAFT_REPO_ROOT
for ab in 0 512 0; do
  CS_EXTRA_HIPCC_FLAGS="-DCS_ABLATE=$ab" python -m commonscenes_amd.build --force > /dev/null 2>&1
  echo "== CS_ABLATE=$ab"
  for i in 0 1 3 5; do python tools/gemm_bench.py --math f16x3 --only $i --iters 10 2>&1 | grep -v amdgpu.ids; done
done
python -m commonscenes_amd.build --force > /dev/null 2>&1
