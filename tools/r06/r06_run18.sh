#!/bin/bash
# per-shape durations of the Winograd transform kernels inside the step loop (rocprofv3 kernel trace, by grid size)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
REPO=$(pwd); cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_t -o bench -- python $REPO/bench.py --no-cpu-baseline --no-extras --no-fp32-leg --no-traffic --no-gemm-profile --steps 10 --warmup 3 > /dev/null 2>&1
cd $REPO
DB=$(find /tmp/prof_t -name "*.db" | head -1)
{ python tools/rocpd_by_grid.py $DB gn_apply_wino43 | grep -v columns; python tools/rocpd_by_grid.py $DB wino_out_kernel | grep -v columns; python tools/rocpd_by_grid.py $DB ln_pair | grep -v columns;  python tools/rocpd_by_grid.py $DB gn_apply_kernel | grep -v columns; } > gpurun_out/r06t_transform_kernels_by_grid.txt
cat gpurun_out/r06t_transform_kernels_by_grid.txt
