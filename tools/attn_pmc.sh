#!/bin/bash
# PMC passes over the eight-wave attention kernels at the UNet's 1024-token shape: tools/attn_pmc.sh <outdir>   (env as given)
set -u
OUT=${1:-gpurun_out/attn_pmc}
REPO=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" \
           "SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$REPO/$OUT/p$i" -- python $REPO/tools/attn_one.py > "$REPO/$OUT/p$i.log" 2>&1 || echo "pass $i failed"
done
cd "$REPO"
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "attn_f16x3" not in r["Kernel_Name"]:
            continue
        tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
with open(out + "/summary.txt", "w") as fh:
    for k in sorted(tot):
        line = f"{k:34s} {tot[k] / max(n[k], 1):16.1f}  (per launch, {n[k]} samples)"
        print(line); fh.write(line + "\n")
PY
rm -rf $OUT/p[0-9]*
