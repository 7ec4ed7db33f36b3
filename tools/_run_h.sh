cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
REPO=$(pwd)
timeout 600 python -m pytest tests/test_parity_depth_gpu.py -m gpu -q -k "overflow or rerun" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
for ob in 7 1; do
timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_h$ob -o bench -- python $REPO/bench.py --no-cpu-baseline --no-traffic --no-extras --objects $ob --steps 10 --warmup 3 --gemm-table > $REPO/gpurun_out/r04_h_bench_c$ob.json 2> $REPO/gpurun_out/r04_h_gemm_table_c$ob.txt
DB=$(find $REPO/gpurun_out/prof_h$ob -name "*.db" | head -1)
[ -n "$DB" ] && python $REPO/tools/rocpd_summary.py $DB > $REPO/gpurun_out/r04_h_c${ob}_kernel_stats.txt
rm -rf $REPO/gpurun_out/prof_h$ob
grep -v "^[WEI]2026" $REPO/gpurun_out/r04_h_gemm_table_c$ob.txt | head -60 > $REPO/gpurun_out/r04_h_gemm_table_c${ob}_clean.txt
done
head -30 $REPO/gpurun_out/r04_h_c7_kernel_stats.txt | cut -c1-150
