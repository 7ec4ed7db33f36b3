#!/bin/bash
# PMC passes over one GEMM shape (tools/gemm_bench.py --only N): where the dominant kernel's cycles go.
# usage (on the GPU box): tools/pmc_gemm.sh <shape-index> <tile> <outdir>
#   PMC_CMD overrides the benchmark command (default: tools/gemm_bench.py on the shape; e.g.
#   PMC_CMD="tools/conv_pre_bench.py --only 3 --iters 3 --pre-only" profiles the slab kernel on pre-split operands)
# Counter passes are separate runs with --kernel-trace only (no other trace domain), as the pool requires.
set -u
SHAPE=${1:-3}; TILE=${2:-0}; OUT=${3:-gpurun_out/pmc}
REPO=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$REPO/$OUT/p$i" -- \
    python $REPO/${PMC_CMD:-tools/gemm_bench.py --math f16x3 --only $SHAPE --tile $TILE --iters 3} > "$REPO/$OUT/p$i.log" 2>&1 || echo "pass $i failed"
done
cd "$REPO"
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv_gemm_f16x3" not in r["Kernel_Name"]:
            continue
        tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
with open(out + "/summary.txt", "w") as fh:
    for k in sorted(tot):
        line = f"{k:34s} {tot[k] / max(n[k], 1):16.1f}  (per launch, {n[k]} samples)"
        print(line); fh.write(line + "\n")
PY
