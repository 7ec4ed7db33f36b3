#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=r05_f
timeout 900 python -m pytest tests/test_kwave_gpu.py tests/test_model_gpu.py::test_vq_decode_batch_invariance tests/test_vqvae_native_gpu.py tests/test_model_gpu.py::test_inference_graph2shape_gen_shape_after_foward -q -s > gpurun_out/${T}_new_tests.log 2>&1; echo "new tests rc=$?" | tee -a gpurun_out/${T}_new_tests.log
grep -v "^$" gpurun_out/${T}_new_tests.log | grep -v Warning | tail -30 | cut -c1-300
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/${T}_tests.log 2>&1; echo "suite rc=$?" | tee -a gpurun_out/${T}_tests.log
tail -12 gpurun_out/${T}_tests.log | cut -c1-220
for rep in 1 2; do
  for v in 1 0; do
    ms=$(CS_ATTN_NW2=$v timeout 600 python bench.py --no-cpu-baseline --no-traffic --no-extras --no-fp32-leg --objects 1 --steps 40 --warmup 5 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "objects=1 CS_ATTN_NW2=$v rep=$rep ms_per_step=$ms" | tee -a gpurun_out/${T}_attn_nw2_ab.txt
  done
done
for rep in 1 2; do
  for v in 0 1; do
    if [ $v = 1 ]; then export CS_NO_GN_PARTS=1; else unset CS_NO_GN_PARTS; fi
    timeout 600 python tools/decode_bench.py 2>/dev/null | head -1 | sed "s/^/CS_NO_GN_PARTS=$v rep=$rep: /" | tee -a gpurun_out/${T}_decode_gn_parts_ab.txt
  done
done
unset CS_NO_GN_PARTS
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_${T}3 -o bench -- python $REPO/bench.py --no-cpu-baseline --no-traffic --no-extras --no-fp32-leg --objects 1 --steps 20 --warmup 3 > $REPO/gpurun_out/${T}_bench_c2_under_rocprof.json 2> /dev/null
cd $REPO
DB=$(find gpurun_out/prof_${T}3 -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB > gpurun_out/${T}_c2_kernel_stats.txt
rm -rf gpurun_out/prof_${T}3
head -16 gpurun_out/${T}_c2_kernel_stats.txt | cut -c1-200
