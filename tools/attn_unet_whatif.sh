# timing-only what-if builds of the in-kernel-split attention kernel at the UNet's level-1 shape (1024 tokens x dh 56, batch 64)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B=commonscenes_amd/build
relink() {
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wall -Wno-unused-function $1 -c commonscenes_amd/csrc/cs_attention_f16x3.hip -o $B/cs_attention_f16x3.o &&
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $B/*.o -o commonscenes_amd/libcommonscenes_hip.so
}
for flags in "" "-DCS_ATTN_WHATIF_NO_STAGE" "-DCS_ATTN_WHATIF_NO_EXP" "-DCS_ATTN_WHATIF_NO_RESCALE" "-DCS_ATTN_WHATIF_NO_STAGE -DCS_ATTN_WHATIF_NO_EXP -DCS_ATTN_WHATIF_NO_RESCALE" ""; do
  relink "$flags" 2>/dev/null || echo "build failed: $flags"
  for rep in 1 2 3; do
    echo "[$flags] $(timeout 300 python tools/attn_bench.py 2>&1 | grep 'N= 1024' | cut -d'|' -f2)"
  done
done | tee gpurun_out/r03_af_attn_unet_whatif.txt
relink ""
