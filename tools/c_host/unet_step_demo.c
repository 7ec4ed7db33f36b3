/*
 * A host with no Python and no torch driving the whole-forward entry point of libcommonscenes_hip.so:
 * the sequence INTEGRATION.md section B.1 describes (plan -> raw parameters -> pack -> context -> step -> fused
 * guidance + DDIM update -> cs_vqvae_decode), on synthetic weights (cs_synth_fill).  It checks what a C caller can check on its own:
 *   - every call returns CS_OK and the outputs are finite;
 *   - two runs produce the same bits (no hidden state, deterministic kernels);
 *   - the guidance-pair entry (cfg_pairs = 1: shared (x, t), contexts [uc; c]) equals the duplicated batch
 *     (cfg_pairs = 0) bit for bit -- samples never mix;
 *   - a too-small workspace is refused with CS_ENOMEM.
 * Build (plain C, gcc):  gcc -std=c99 -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I include \
 *     tools/c_host/unet_step_demo.c -L commonscenes_amd -lcommonscenes_hip -L/opt/rocm/lib -lamdhip64 -lm \
 *     -Wl,-rpath,$PWD/commonscenes_amd -Wl,-rpath,/opt/rocm/lib -o /tmp/unet_step_demo
 * (tests/test_c_host_gpu.py does exactly that on the MI355X.)
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "commonscenes_hip.h"

#define CHECK(call)                                                          \
  do {                                                                       \
    int rc_ = (int)(call);                                                   \
    if (rc_ != 0) {                                                          \
      fprintf(stderr, "%s:%d: %s -> %d\n", __FILE__, __LINE__, #call, rc_); \
      return 1;                                                              \
    }                                                                        \
  } while (0)

static uint64_t fnv1a(const char* s) {
  uint64_t h = 1469598103934665603ull;
  for (; *s; ++s) h = (h ^ (uint8_t)*s) * 1099511628211ull;
  return h;
}

static int all_finite(const float* v, size_t n) {
  for (size_t i = 0; i < n; ++i)
    if (!isfinite(v[i])) return 0;
  return 1;
}

int main(int argc, char** argv) {
  const int width = argc > 1 ? atoi(argv[1]) : 32;      /* model_channels: 32 = reduced, 224 = shipped */
  const int B = argc > 2 ? atoi(argv[2]) : 2;           /* objects */
  CsUnetConfig cfg;
  memset(&cfg, 0, sizeof(cfg));
  cfg.in_channels = 3; cfg.out_channels = 3; cfg.model_channels = width; cfg.num_res_blocks = 2;
  cfg.n_mult = 3; cfg.channel_mult[0] = 1; cfg.channel_mult[1] = 2; cfg.channel_mult[2] = 3;
  cfg.n_attn_res = 2; cfg.attention_resolutions[0] = 4; cfg.attention_resolutions[1] = 2;
  cfg.num_heads = 8; cfg.context_dim = 1280; cfg.d = cfg.h = cfg.w = 16;
  cfg.math = CS_MATH_F16X3; cfg.use_spatial_transformer = 1; cfg.dims = 3;

  if (cs_abi_version() < 6) { fprintf(stderr, "library ABI %d too old\n", cs_abi_version()); return 1; }
  cs_unet* u = NULL;
  CHECK(cs_unet_create(&cfg, &u));
  hipStream_t st;
  CHECK(hipStreamCreate(&st));

  /* raw parameters: fan-in scaled uniform noise per tensor, gamma-like vectors around 1 */
  void *raw = NULL, *arena = NULL;
  CHECK(hipMalloc(&raw, (size_t)cs_unet_raw_bytes(u)));
  CHECK(hipMalloc(&arena, (size_t)cs_unet_arena_bytes(u)));
  const int np = cs_unet_param_count(u);
  int64_t total = 0;
  for (int i = 0; i < np; ++i) {
    const char* name; int64_t shape[5], off; int nd;
    CHECK(cs_unet_param_info(u, i, &name, shape, &nd, &off));
    int64_t n = 1, fan_in = 1;
    for (int k = 0; k < nd; ++k) n *= shape[k];
    for (int k = 1; k < nd; ++k) fan_in *= shape[k];
    const int is_norm_w = nd == 1 && strstr(name, "norm") && strstr(name, ".weight");
    const int is_gn_w = nd == 1 && (strstr(name, "in_layers.0.weight") || strstr(name, "out_layers.0.weight") ||
                                    strstr(name, "out.0.weight"));
    const double scale = nd == 1 ? ((is_norm_w || is_gn_w) ? 0.2 : 0.1) : sqrt(3.0 / (double)fan_in);
    const double offset = (is_norm_w || is_gn_w) ? 1.0 : 0.0;
    CHECK(cs_synth_fill((float*)((char*)raw + off), n, fnv1a(name), scale, offset, st));
    total += n;
  }
  CHECK(cs_unet_pack(u, raw, arena, st));
  CHECK(hipStreamSynchronize(st));
  CHECK(hipFree(raw));
  printf("plan: %d tensors, %lld parameters, arena %.1f MB\n", np, (long long)total, cs_unet_arena_bytes(u) / 1e6);

  /* inputs */
  const int S = 16 * 16 * 16, per = 3 * S;
  const int64_t cf = cs_unet_context_floats(u);
  float *x, *x2, *ctx, *ctxvec, *eps, *eps_dup, *xprev;
  int64_t *t, *t2;
  CHECK(hipMalloc((void**)&x, sizeof(float) * B * per));
  CHECK(hipMalloc((void**)&x2, sizeof(float) * 2 * B * per));
  CHECK(hipMalloc((void**)&xprev, sizeof(float) * B * per));
  CHECK(hipMalloc((void**)&ctx, sizeof(float) * 2 * B * 1280));
  CHECK(hipMalloc((void**)&ctxvec, sizeof(float) * 2 * B * cf));
  CHECK(hipMalloc((void**)&eps, sizeof(float) * 2 * B * per));
  CHECK(hipMalloc((void**)&eps_dup, sizeof(float) * 2 * B * per));
  CHECK(hipMalloc((void**)&t, sizeof(int64_t) * B));
  CHECK(hipMalloc((void**)&t2, sizeof(int64_t) * 2 * B));
  CHECK(cs_synth_fill(x, (int64_t)B * per, 11, 1.7, 0.0, st));
  CHECK(cs_synth_fill(ctx, (int64_t)2 * B * 1280, 12, 1.7, 0.0, st));
  int64_t* th = (int64_t*)malloc(sizeof(int64_t) * 2 * B);
  for (int i = 0; i < 2 * B; ++i) th[i] = 981;
  CHECK(hipMemcpyAsync(t, th, sizeof(int64_t) * B, hipMemcpyHostToDevice, st));
  CHECK(hipMemcpyAsync(t2, th, sizeof(int64_t) * 2 * B, hipMemcpyHostToDevice, st));
  CHECK(hipMemcpyAsync(x2, x, sizeof(float) * B * per, hipMemcpyDeviceToDevice, st));
  CHECK(hipMemcpyAsync(x2 + (size_t)B * per, x, sizeof(float) * B * per, hipMemcpyDeviceToDevice, st));

  int64_t wsb = cs_unet_workspace_bytes(u, B, 1);
  const int64_t wsb_dup = cs_unet_workspace_bytes(u, 2 * B, 0);
  if (wsb_dup > wsb) wsb = wsb_dup;
  if (wsb <= 0) { fprintf(stderr, "workspace_bytes -> %lld\n", (long long)wsb); return 1; }
  void* ws;
  CHECK(hipMalloc(&ws, (size_t)wsb));
  int32_t* status;                 /* sticky CS_STATUS_* word: F16X3 kernels report fp16-range overflow here */
  CHECK(hipMalloc((void**)&status, sizeof(int32_t)));
  CHECK(hipMemsetAsync(status, 0, sizeof(int32_t), st));
  CHECK(cs_unet_context(u, arena, ctx, 2 * B, ctxvec, status, ws, wsb, st));

  if (cs_unet_step(u, arena, x, t, ctxvec, eps, B, 1, status, ws, 4096, st) != CS_ENOMEM) {
    fprintf(stderr, "a 4 KB workspace was not refused\n");
    return 1;
  }
  const size_t ne = (size_t)2 * B * per;
  float *h1 = (float*)malloc(sizeof(float) * ne), *h2 = (float*)malloc(sizeof(float) * ne);
  CHECK(cs_unet_step(u, arena, x, t, ctxvec, eps, B, 1, status, ws, wsb, st));
  CHECK(hipMemcpyAsync(h1, eps, sizeof(float) * ne, hipMemcpyDeviceToHost, st));
  CHECK(cs_unet_step(u, arena, x, t, ctxvec, eps, B, 1, status, ws, wsb, st));
  CHECK(hipMemcpyAsync(h2, eps, sizeof(float) * ne, hipMemcpyDeviceToHost, st));
  CHECK(hipStreamSynchronize(st));
  if (!all_finite(h1, ne)) { fprintf(stderr, "non-finite eps\n"); return 1; }
  if (memcmp(h1, h2, sizeof(float) * ne)) { fprintf(stderr, "two runs differ\n"); return 1; }
  CHECK(cs_unet_step(u, arena, x2, t2, ctxvec, eps_dup, 2 * B, 0, status, ws, wsb, st));
  CHECK(hipMemcpyAsync(h2, eps_dup, sizeof(float) * ne, hipMemcpyDeviceToHost, st));
  CHECK(hipStreamSynchronize(st));
  double num = 0, den = 0;
  for (size_t i = 0; i < ne; ++i) { const double d = (double)h1[i] - h2[i]; num += d * d; den += (double)h2[i] * h2[i]; }
  const double rel = sqrt(num / den);
  printf("guidance-pair entry vs duplicated batch: rel-L2 %.3e%s\n", rel, rel == 0.0 ? " (bit-identical)" : "");
  if (rel > 1e-5) { fprintf(stderr, "cfg_pairs result differs from the duplicated batch\n"); return 1; }

  /* one fused guidance + DDIM update on top (samplers/ddim.py:206-243) */
  CHECK(cs_ddim_cfg_update(x, eps, NULL, xprev, NULL, B, per, 0.0047f, 0.0058f, 0.f, 0.99765f, 3.0f, 1, st));
  CHECK(hipMemcpyAsync(h1, xprev, sizeof(float) * B * per, hipMemcpyDeviceToHost, st));
  CHECK(hipStreamSynchronize(st));
  if (!all_finite(h1, (size_t)B * per)) { fprintf(stderr, "non-finite x_prev\n"); return 1; }
  /* ... and the downstream end: VQVAE.decode_no_quant of the updated latents (config/vqvae_snet.yaml) */
  CsVqvaeConfig vc;
  memset(&vc, 0, sizeof(vc));
  vc.ch = 64; vc.out_ch = 1; vc.n_mult = 3; vc.ch_mult[0] = 1; vc.ch_mult[1] = 2; vc.ch_mult[2] = 4;
  vc.num_res_blocks = 1; vc.z_channels = 3; vc.resolution = 64; vc.n_embed = 8192; vc.embed_dim = 3;
  vc.math = CS_MATH_F16X3;
  cs_vqvae* vq = NULL;
  CHECK(cs_vqvae_create(&vc, &vq));
  void *vraw = NULL, *varena = NULL, *vws = NULL;
  CHECK(hipMalloc(&vraw, (size_t)cs_vqvae_raw_bytes(vq)));
  CHECK(hipMalloc(&varena, (size_t)cs_vqvae_arena_bytes(vq)));
  for (int i = 0; i < cs_vqvae_param_count(vq); ++i) {
    const char* name; int64_t shape[5], off; int nd;
    CHECK(cs_vqvae_param_info(vq, i, &name, shape, &nd, &off));
    int64_t n = 1, fan_in = 1;
    for (int k = 0; k < nd; ++k) n *= shape[k];
    for (int k = 1; k < nd; ++k) fan_in *= shape[k];
    const int gamma = nd == 1 && strstr(name, "norm") && strstr(name, ".weight");
    const int book = strstr(name, "embedding") != NULL;
    CHECK(cs_synth_fill((float*)((char*)vraw + off), n, fnv1a(name), book ? 1.5 : (nd == 1 ? (gamma ? 0.2 : 0.1) : sqrt(3.0 / (double)fan_in)),
                        gamma ? 1.0 : 0.0, st));
  }
  CHECK(cs_vqvae_pack(vq, vraw, varena, st));
  const int64_t vwsb = cs_vqvae_workspace_bytes(vq, B);
  if (vwsb <= 0) { fprintf(stderr, "cs_vqvae_workspace_bytes -> %lld\n", (long long)vwsb); return 1; }
  CHECK(hipMalloc(&vws, (size_t)vwsb));
  float* sdf;
  const size_t nsdf = (size_t)B * 64 * 64 * 64;
  CHECK(hipMalloc((void**)&sdf, sizeof(float) * nsdf));
  CHECK(cs_vqvae_decode(vq, varena, xprev, sdf, NULL, B, 1, status, vws, vwsb, st));
  float* hs = (float*)malloc(sizeof(float) * nsdf);
  CHECK(hipMemcpyAsync(hs, sdf, sizeof(float) * nsdf, hipMemcpyDeviceToHost, st));
  CHECK(hipStreamSynchronize(st));
  if (!all_finite(hs, nsdf)) { fprintf(stderr, "non-finite SDF\n"); return 1; }
  int32_t hstatus = -1;
  CHECK(hipMemcpy(&hstatus, status, sizeof(int32_t), hipMemcpyDeviceToHost));
  if (hstatus != 0) { fprintf(stderr, "status word %d (CS_STATUS_F16X3_OVERFLOW?)\n", (int)hstatus); return 1; }
  printf("decode: %d x 64^3 SDF, workspace %.1f MB\n", B, vwsb / 1e6);
  cs_vqvae_destroy(vq);

  double rms = 0;
  for (size_t i = 0; i < ne; ++i) rms += (double)h2[i] * h2[i];
  printf("ok: width %d, %d objects, eps rms %.4f, workspace %.1f MB\n", width, B, sqrt(rms / ne), wsb / 1e6);
  cs_unet_destroy(u);
  return 0;
}
