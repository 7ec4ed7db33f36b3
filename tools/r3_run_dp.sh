# round-3: rocprofv3 kernel stats of the VQ-VAE decode alone (32 objects, 4 decodes)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_r03dp -o dec -- python $REPO/tools/decode_bench.py > $REPO/gpurun_out/r03_u_decode_table.txt 2>&1
cd $REPO
DB=$(find gpurun_out/prof_r03dp -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB > gpurun_out/r03_u_decode_kernel_stats.txt
rm -rf gpurun_out/prof_r03dp
head -40 gpurun_out/r03_u_decode_kernel_stats.txt
cat gpurun_out/r03_u_decode_table.txt | grep -v amdgpu
