# A/B of the split-K plan on one box: round-1 rule vs the round-2 rule, ms/step at 1 / 4 / 7 objects
cd $GRAFT_REPO_ROOT
for flag in "-DCS_PLAN_R1" ""; do
  CS_EXTRA_HIPCC_FLAGS="$flag" python -m commonscenes_amd.build --force > /dev/null 2>&1
  for rep in 1 2; do for B in 1 4 7; do
    python bench.py --no-cpu-baseline --no-extras --objects $B --steps 30 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('plan[$flag] B=$B', round(d['ms_per_step'],3))"
  done; done
done
python -m commonscenes_amd.build --force > /dev/null 2>&1
