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
// exact GELU (attention.py:44-46 F.gelu, vqvae_modules.py: nn.GELU): 0.5 x (1 + erf(x / sqrt 2)).
// (r6, measured and not kept: a hand-rolled erf -- the device library's two minimax branches with v_exp_f32 in place of its
// range-reduced exp, 25 instead of 45 instructions per GELU, <= 1.5 ulp -- on the suspicion that the fused GEGLU epilogue was
// VALU-bound on erff: same-box A/B of the whole step 62.93 / 62.82 vs 62.90 / 62.86 ms, i.e. nothing; the library's erff stays.
// profiles/r06_c_erf_whatif_ab.txt)
__device__ __forceinline__ float cs_gelu(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
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
