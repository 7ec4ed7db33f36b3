cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B=commonscenes_amd/build
relink() {
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wall -Wno-unused-function $1 -c commonscenes_amd/csrc/cs_attention_f16x3.hip -o $B/cs_attention_f16x3.o &&
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $B/*.o -o commonscenes_amd/libcommonscenes_hip.so
}
for flags in "" "-DCS_ATTN_WHATIF_NO_DMA" "-DCS_ATTN_WHATIF_NO_BARRIER" "-DCS_ATTN_WHATIF_NO_LDSREAD" "-DCS_ATTN_WHATIF_NO_LDSREAD -DCS_ATTN_WHATIF_NO_DMA -DCS_ATTN_WHATIF_NO_BARRIER" "-DCS_ATTN_WHATIF_NO_LDSREAD -DCS_ATTN_WHATIF_NO_DMA -DCS_ATTN_WHATIF_NO_BARRIER -DCS_ATTN_WHATIF_NO_RESCALE -DCS_ATTN_WHATIF_NO_EXP" ""; do
  relink "$flags" 2>/dev/null || echo "build failed: $flags"
  for rep in 1 2; do
    echo "[$flags] $(timeout 300 python tools/attn_bench.py 2>&1 | grep 'N= 4096' | cut -d'|' -f2)"
  done
done | tee gpurun_out/r03_ad_attn256_whatif2.txt
