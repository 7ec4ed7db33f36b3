# What-if on the slab kernel fed pre-split operands: CS_ABLATE=8 drops the per-chunk s_barrier, 12 the vmcnt wait too
# (racy: results wrong, timing only) -- the ceiling of anything that synchronises less often.
cd $GRAFT_REPO_ROOT
for ab in 0 8 12 0; do
  CS_EXTRA_HIPCC_FLAGS="-DCS_ABLATE=$ab" python -m commonscenes_amd.build --force > /dev/null 2>&1
  echo "== CS_ABLATE=$ab"
  python tools/conv_pre_bench.py --pre-only 2>&1 | grep -v amdgpu.ids
done
python -m commonscenes_amd.build --force > /dev/null 2>&1
