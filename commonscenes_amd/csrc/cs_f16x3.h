// Shared pieces of the CS_MATH_F16X3 GEMM kernels (cs_gemm_f16x3.hip, cs_gemm_kw.hip).
#pragma once
#include "cs_common.h"

// r5: the split-K reduce + epilogue folded into the slice kernel (CsConvGemm.splitk_sync; cs_gemm.hip decides).  n = 0: off.
// The final descriptor's epilogue terms travel here; the kernel's own descriptor `p` stays the plain partial-tile one.
struct CsFuseK {
  float* out;
  const float* bias;
  const float* scale;
  const float* shift;
  const float* rowvec;
  const float* res;
  double* gn_part;
  int32_t* sync;          // two zeroed words per output tile: arrivals, departures (returned to zero by the last reducer)
  int32_t* status;
  int32_t ldo, ldr, ldrv, rv_rows, act, gn_ld, out_format;
  int32_t reducers;       // slices 0 .. reducers - 1 of a tile each reduce BM / reducers rows (a multiple of 16)
  float out_scale;
};


namespace cs16 {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

constexpr int BKH = 16;                 // K elements per chunk
constexpr float A_SCALE_DEFAULT = 16.0f;
constexpr unsigned OOB = 0xFFF00000u;   // byte offset past every buffer (extents are < 0xFFE00000): reads 0

// `amax` is the lane's running max |a * a_scale|: the kernel raises CS_STATUS_F16X3_OVERFLOW when it reaches the fp16
// range (the hi half would be +-inf).  Four v_max3_f32 per eight elements, hidden under the MFMA stream.
__device__ __forceinline__ void split8(const f32x4& x, const f32x4& y, float a_scale, h8& hi, h8& lo, float& amax) {
  const float v[8] = {x[0] * a_scale, x[1] * a_scale, x[2] * a_scale, x[3] * a_scale,
                      y[0] * a_scale, y[1] * a_scale, y[2] * a_scale, y[3] * a_scale};
  amax = fmaxf(fmaxf(amax, fabsf(v[0])), fabsf(v[1]));
  amax = fmaxf(fmaxf(amax, fabsf(v[2])), fabsf(v[3]));
  amax = fmaxf(fmaxf(amax, fabsf(v[4])), fabsf(v[5]));
  amax = fmaxf(fmaxf(amax, fabsf(v[6])), fabsf(v[7]));
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const _Float16 h = (_Float16)v[i];
    hi[i] = h;
    lo[i] = (_Float16)(v[i] - (float)h);
  }
}

// split8 with a per-lane scale that may be 0 (a masked-out conv tap): the scaling multiply follows the "legacy" rule
// 0 * x = 0 for EVERY x, so a NaN or Inf sitting in a masked slab row (another sample's data) cannot leak through.
// For a non-zero scale the product is the IEEE one: results equal split8's.
__device__ __forceinline__ void split8_masked(const f32x4& x, const f32x4& y, float sc, h8& hi, h8& lo, float& amax) {
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float in = i < 4 ? x[i] : y[i - 4];
    asm("v_mul_legacy_f32 %0, %1, %2" : "=v"(v[i]) : "v"(in), "v"(sc));
  }
  amax = fmaxf(fmaxf(amax, fabsf(v[0])), fabsf(v[1]));
  amax = fmaxf(fmaxf(amax, fabsf(v[2])), fabsf(v[3]));
  amax = fmaxf(fmaxf(amax, fabsf(v[4])), fabsf(v[5]));
  amax = fmaxf(fmaxf(amax, fabsf(v[6])), fabsf(v[7]));
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const _Float16 h = (_Float16)v[i];
    hi[i] = h;
    lo[i] = (_Float16)(v[i] - (float)h);
  }
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  // (the immediate is a template constant: "n" operand of the inline assembly)
  static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit field");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

}  // namespace cs16
