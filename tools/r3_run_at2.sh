# round-3: the 256-wide attention variant -- two QK accumulator chains / exact rescale skip, A/B builds on one box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B=commonscenes_amd/build
relink() {
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wall -Wno-unused-function $1 -c commonscenes_amd/csrc/cs_attention_f16x3.hip -o $B/cs_attention_f16x3.o &&
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $B/*.o -o commonscenes_amd/libcommonscenes_hip.so
}
for flags in "-DCS_ATTN_ONE_CHAIN -DCS_ATTN_ALWAYS_RESCALE" "-DCS_ATTN_ALWAYS_RESCALE" "-DCS_ATTN_ONE_CHAIN" ""; do
  relink "$flags" || echo "build failed: $flags"
  for rep in 1 2; do
    echo "[$flags] $(timeout 300 python tools/attn_bench.py 2>&1 | grep 'N= 4096')"
  done
done | tee gpurun_out/r03_w_attn256_ab.txt
( timeout 900 python -m pytest tests/test_f16x3_gpu.py tests/test_vqvae_native_gpu.py tests/test_model_gpu.py -m gpu -q -x > gpurun_out/r03_w_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_w_tests.log )
tail -5 gpurun_out/r03_w_tests.log
timeout 300 python tools/decode_bench.py 2>&1 | grep -v amdgpu.ids | head -1
