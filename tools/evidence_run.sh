#!/bin/bash
# Evidence run on the GPU box (one script for every round; the per-round one-off scripts of r3 are gone):
#   tools/evidence_run.sh <tag> [quick]
# suite, default bench line (+ live traffic and MFMA-busy PMC passes), rocprofv3 --kernel-trace --stats summaries of the
# default command with the GEMM table and of the step loop alone; without `quick` also the HBM-bound kernels and decode tables.
# Everything lands in gpurun_out/<tag>_*; copy what is to be judged into profiles/.
TAG=${1:-r04_a}; MODE=${2:-full}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( CS_PARITY_LOG=$PWD/gpurun_out/${TAG}_parity_log.txt timeout 1500 python -m pytest tests -m gpu -q -x --durations=8 > gpurun_out/${TAG}_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_tests.log )
tail -14 gpurun_out/${TAG}_tests.log
timeout 1200 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"
tail -3 gpurun_out/${TAG}_bench.err
python - <<PY
import json
try:
    d = json.load(open('gpurun_out/${TAG}_bench.json')); r = d['roofline']
    print('steps/s', d['value'], 'ms', d['ms_per_step'], 'frac', r['frac'], 'avg_ms', r['avg_launch_ms'], 'traffic', r['traffic'],
          'busy', r.get('mfma_busy_frac'), 'clk', r.get('effective_clock_ghz'))
    print('c2', d['c2']['ms_per_step'], 'c7', d['c7']['ms_per_step'], d['c7']['ms_per_object_step'], 'decode/obj', d['decode']['ms_per_object'])
    print('c7x5', json.dumps(d.get('c7x5')))
    print('cpu', d['cpu_baseline']['value'], 'fp32', d['fp32_mfma']['value'])
    print('busy detail', json.dumps(r.get('mfma_busy_detail')))
except Exception as e:
    print('no bench line', e)
PY
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_${TAG} -o bench -- python $REPO/bench.py --no-cpu-baseline --no-traffic --gemm-table > $REPO/gpurun_out/${TAG}_bench_under_rocprof.json 2> $REPO/gpurun_out/${TAG}_gemm_table.txt
timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_${TAG}2 -o bench -- python $REPO/bench.py --no-cpu-baseline --no-extras > $REPO/gpurun_out/${TAG}_bench_steploop_under_rocprof.json 2> /dev/null
timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_${TAG}3 -o bench -- python $REPO/bench.py --no-cpu-baseline --no-extras --objects 1 --steps 20 --warmup 3 > $REPO/gpurun_out/${TAG}_bench_c2_under_rocprof.json 2> /dev/null
timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_${TAG}4 -o bench -- python $REPO/bench.py --no-cpu-baseline --no-traffic --no-extras --objects 7 --steps 10 --warmup 3 --gemm-table > $REPO/gpurun_out/${TAG}_bench_c7_under_rocprof.json 2> $REPO/gpurun_out/${TAG}_gemm_table_c7.txt
cd $REPO
DB=$(find gpurun_out/prof_${TAG}4 -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB > gpurun_out/${TAG}_c7_kernel_stats.txt
grep -v "^[WEI]2026" gpurun_out/${TAG}_gemm_table_c7.txt | head -45 > gpurun_out/${TAG}_gemm_table_c7_clean.txt
DB=$(find gpurun_out/prof_${TAG} -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB > gpurun_out/${TAG}_kernel_stats.txt
DB=$(find gpurun_out/prof_${TAG}2 -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB > gpurun_out/${TAG}_kernel_stats_steploop.txt && python tools/rocpd_by_grid.py $DB "conv_gemm_f16x3_kernel<1, 7, 8, 1, true, 32," > gpurun_out/${TAG}_dominant_kernel_by_grid.txt && head -8 gpurun_out/${TAG}_kernel_stats_steploop.txt | cut -c1-170
DB=$(find gpurun_out/prof_${TAG}3 -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB > gpurun_out/${TAG}_c2_kernel_stats.txt
rm -rf gpurun_out/prof_${TAG} gpurun_out/prof_${TAG}2 gpurun_out/prof_${TAG}3 gpurun_out/prof_${TAG}4
grep -v "^[WEI]2026" gpurun_out/${TAG}_gemm_table.txt | head -45 > gpurun_out/${TAG}_gemm_table_clean.txt
if [ "$MODE" != "quick" ]; then
  python tools/hbm_bench.py > gpurun_out/${TAG}_hbm_bound_kernels.txt 2>&1
  python tools/decode_bench.py > gpurun_out/${TAG}_decode_table.txt 2>&1
  for nb in 2 14 64; do SM_BATCH=$nb timeout 600 python tools/gemm_tok_smallm.py > gpurun_out/${TAG}_tok_smallm_b$nb.txt 2>&1; done
  SM_BATCH=2 timeout 600 python tools/gemm_smallm_tiles.py > gpurun_out/${TAG}_smallm_tiles_b2.txt 2>&1
  head -3 gpurun_out/${TAG}_decode_table.txt
fi
