# What-if on the token (1-tap) GEMMs: CS_ABLATE=64 serves every operand fetch from one 16 KB window (no HBM / L2 misses on
# the load side; stores unchanged), CS_ABLATE=32 takes the scalar epilogue out of the picture -- timing only.
cd $GRAFT_REPO_ROOT
for ab in 0 64; do
  CS_EXTRA_HIPCC_FLAGS="-DCS_ABLATE=$ab" python -m commonscenes_amd.build --force > /dev/null 2>&1
  echo "== CS_ABLATE=$ab"
  python tools/gemm_1tap.py 2>&1 | grep -v amdgpu.ids | cut -c1-70
done
python -m commonscenes_amd.build --force > /dev/null 2>&1
