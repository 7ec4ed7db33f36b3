// Shared device helpers for libcommonscenes_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/commonscenes_hip.h"

#define CS_CHECK_LAUNCH()                         \
  do {                                            \
    hipError_t e_ = hipGetLastError();            \
    if (e_ != hipSuccess) return (int)e_;         \
  } while (0)

// hipGetLastError() is sticky per thread: clear whatever an earlier, unrelated runtime call left behind
// (e.g. a device probe) so CS_CHECK_LAUNCH reports this launch only.
#define CS_LAUNCH(...)              \
  do {                              \
    (void)hipGetLastError();        \
    hipLaunchKernelGGL(__VA_ARGS__); \
  } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float cs_silu(float x) { return x / (1.0f + __expf(-x)); }
// exact SiLU: x * sigmoid(x) with an accurate exp (torch computes x * 1/(1+exp(-x)) in fp32)
#ifdef CS_FAST_SILU      // what-if build (tools/build_alt.sh): v_exp_f32 + v_rcp_f32 instead of the accurate exp + IEEE division (~2.5 ulp)
__device__ __forceinline__ float cs_silu_acc(float x) {
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -0x1.715476p+0f));
}
#else
__device__ __forceinline__ float cs_silu_acc(float x) { return x / (1.0f + expf(-x)); }
#endif
// erf(a) in fp32, <= 1.5 ulp (r6).  The device library's erff -- the same two minimax branches, whose coefficients these are --
// spends twelve of its ~45 instructions on a range-reduced exp and its overflow selects; for the branch |a| >= 1 the result
// is 1 - exp(-r) with r >= 1.2, where v_exp_f32 (1 ulp of an exp <= 0.3, argument rounding ~|r| 2^-24 relative) already sits
// below half an ulp of the result.  The fused GEGLU epilogue of ff.net.0.proj evaluates 56 exact GELUs per lane and tile and
// was VALU-bound on them (epilogue-only timing 247 us of the 842 us 448 -> 3584 GEMM at batch 64, profiles/r05_m_tok_phase.txt).
__device__ __forceinline__ float cs_erff(float a) {
  const float t = fabsf(a);
  float r;
  if (t >= 1.0f) {
    r = fmaf(t, 0x1.1d3156p-16f, -0x1.8d129p-12f);
    r = fmaf(t, r, 0x1.f9a6d2p-9f);
    r = fmaf(t, r, -0x1.8c3164p-6f);
    r = fmaf(t, r, 0x1.b4e9c8p-4f);
    r = fmaf(t, r, 0x1.4515fap-1f);
    r = fmaf(t, r, 0x1.078e5p-3f);
    r = fmaf(t, r, t);
    r = 1.0f - __builtin_amdgcn_exp2f(r * -0x1.715476p+0f);       // exp(-r); -> 1 for r beyond ~17 (exp2 underflows to 0)
  } else {
    const float s = a * a;
    r = fmaf(s, -0x1.268bc2p-11f, 0x1.420828p-8f);
    r = fmaf(s, r, -0x1.b5937p-6f);
    r = fmaf(s, r, 0x1.ce077cp-4f);
    r = fmaf(s, r, -0x1.81266p-2f);
    r = fmaf(s, r, 0x1.06ebap-3f);
    r = fmaf(t, r, t);
  }
  return copysignf(r, a);
}
// exact GELU (attention.py:44-46 F.gelu, vqvae_modules.py: nn.GELU): 0.5 x (1 + erf(x / sqrt 2))
__device__ __forceinline__ float cs_gelu(float x) {
#ifdef CS_OCML_ERF       // what-if build (tools/build_alt.sh): the device library's erff, as until r5 (same-box A/B of cs_erff)
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
#else
  return 0.5f * x * (1.0f + cs_erff(x * 0.70710678118654752440f));
#endif
}
__device__ __forceinline__ float cs_act(float v, int act) {
  switch (act) {
    case CS_ACT_RELU: return v > 0.f ? v : 0.f;
    case CS_ACT_SILU: return cs_silu_acc(v);
    case CS_ACT_GELU: return cs_gelu(v);
    default: return v;
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

static inline int cs_grid_for(int64_t n, int block, int cap = 256 * 16) {
  int64_t g = (n + block - 1) / block;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}
