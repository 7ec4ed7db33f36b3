# A/B on one box at small batches (1 / 4 / 7 objects): flags given as arguments vs the default build
cd $GRAFT_REPO_ROOT
for flag in "$1" ""; do
  CS_EXTRA_HIPCC_FLAGS="$flag" python -m commonscenes_amd.build --force > /dev/null 2>&1
  for rep in 1 2; do for B in 1 4 7; do
    python bench.py --no-cpu-baseline --no-extras --objects $B --steps 30 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$flag] B=$B', round(d['ms_per_step'],3))"
  done; done
done
python -m commonscenes_amd.build --force > /dev/null 2>&1
