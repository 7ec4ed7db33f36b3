#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('SMOKE OK')" 2>&1 | tail -2
for f in "" "--batch-scenes"; do
  python tools/eval_walkthrough.py --scenes 8 --width 224 --ddim-steps 100 $f 2>/dev/null | grep EVAL_WALKTHROUGH > /tmp/w.txt
  python - "$f" <<'PY' | tee -a gpurun_out/r05_walkthrough_wino.txt
import sys, json
d = json.loads(open('/tmp/w.txt').read().split("EVAL_WALKTHROUGH ", 1)[1])
print("flag=[%s]" % sys.argv[1], {k: v for k, v in d.items() if k != "scenes"})
PY
done
