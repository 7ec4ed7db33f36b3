set -e
cd $GRAFT_REPO_ROOT
for ab in 0 1 2 3 4 8 16 19; do
  CS_EXTRA_HIPCC_FLAGS="-DPW_ABLATE=$ab" python -m commonscenes_amd.build --force > /dev/null 2>&1
  echo "== PW_ABLATE=$ab"
  python tools/gemm_1tap.py 2>&1 | awk '{print $1,$2,$3,$4,$(NF-6),$(NF-5),$(NF-4),$(NF-3),$(NF-2)}' | sed -n '1p;3p;4p'
done
python -m commonscenes_amd.build --force > /dev/null 2>&1
