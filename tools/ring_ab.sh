# A/B on one box: ring depth of the 64x64 tile (3 stages vs 6) and the single-launch GroupNorm, ms/step at 1 / 2 / 4 objects
cd $GRAFT_REPO_ROOT
for flag in "-DCS_RING3" ""; do
  CS_EXTRA_HIPCC_FLAGS="$flag" python -m commonscenes_amd.build --force > /dev/null 2>&1
  for rep in 1 2; do for B in 1 2 4; do
    for gn in 0 4096; do
      CS_GN_SMALL_GROUP=$gn python bench.py --no-cpu-baseline --no-extras --objects $B --steps 30 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ring[$flag] gn_small=$gn B=$B', round(d['ms_per_step'],3))"
    done
  done; done
done
python -m commonscenes_amd.build --force > /dev/null 2>&1
