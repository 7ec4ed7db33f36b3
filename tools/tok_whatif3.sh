# Where a token GEMM's time goes (r3): timing-only builds without the epilogue (CS_ABLATE=1024) and without the K loop
# (2048), against the product kernel, same box; tile 4 (256x224, one workgroup per CU) and tile 2 (128x224, two per CU)
cd $GRAFT_REPO_ROOT
for ab in 0 1024 2048; do
  CS_EXTRA_HIPCC_FLAGS="-DCS_ABLATE=$ab" python -m commonscenes_amd.build --force > /dev/null 2>&1
  echo "== CS_ABLATE=$ab"
  python tools/gemm_1tap.py 2>&1 | grep -v amdgpu.ids | cut -c1-105
done
python -m commonscenes_amd.build --force > /dev/null 2>&1
