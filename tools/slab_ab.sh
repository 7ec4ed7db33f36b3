# A/B on one box: per-tap A gather (-DCS_NO_SLAB) vs the shared slab, default bench (32 objects) and decode
cd $GRAFT_REPO_ROOT
for flag in "-DCS_NO_SLAB" ""; do
  CS_EXTRA_HIPCC_FLAGS="$flag" python -m commonscenes_amd.build --force > /dev/null 2>&1
  for rep in 1 2; do
    python bench.py --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('slab[$flag]', 'ms/step', round(d['ms_per_step'],2), 'frac', round(d['roofline']['frac'],4), 'decode ms', round(d['decode']['ms'],1), 'c2', round(d['c2']['ms_per_step'],2))"
  done
done
python -m commonscenes_amd.build --force > /dev/null 2>&1
