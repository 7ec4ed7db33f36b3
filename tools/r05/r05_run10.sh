#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=r05_j
for o in 7 14; do
for rep in 1 2; do
  for v in 1024 2048 4096; do
    ms=$(CS_KWAVE_MAX_TILES=$v timeout 600 python bench.py --no-cpu-baseline --no-traffic --no-extras --no-fp32-leg --no-gemm-profile --objects $o --steps 40 --warmup 5 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "objects=$o CS_KWAVE_MAX_TILES=$v rep=$rep ms_per_step=$ms" | tee -a gpurun_out/${T}_kwave_tiles_ab.txt
  done
done
done
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_${T}4 -o bench -- python $REPO/bench.py --no-cpu-baseline --no-traffic --no-extras --no-fp32-leg --objects 7 --steps 10 --warmup 3 --gemm-table > $REPO/gpurun_out/${T}_bench_c7_under_rocprof.json 2> $REPO/gpurun_out/${T}_gemm_table_c7.txt
cd $REPO
DB=$(find gpurun_out/prof_${T}4 -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB > gpurun_out/${T}_c7_kernel_stats.txt
rm -rf gpurun_out/prof_${T}4
head -28 gpurun_out/${T}_c7_kernel_stats.txt | cut -c1-180
grep -v "^[WEI]2026" gpurun_out/${T}_gemm_table_c7.txt | head -42
