// Flash attention with fp32 operands carried as fp16 hi/lo pairs on the fp16 MFMA (CS_MATH_F16X3 companion of
// cs_attention.hip; same "swapped" formulation, same contract, ~fp32 accuracy at a 5x higher pipe ceiling).
//
//   S^T[j][i] = sum_d K[j][d] Q[i][d]   : A = K tile  (LDS, [key][d] fp16 hi/lo images), B = Q (registers, hi/lo)
//   O^T[d][i] += sum_j V[j][d] P^T[j][i] : A = V^T tile (LDS, [d][key'] fp16 hi/lo images), B = P^T straight from the
//                                          S^T accumulator registers (split into hi/lo in place)
// each contraction = 3 v_mfma_f32_32x32x16_f16 (lo*hi + hi*lo + hi*hi), fp32 accumulate.
//
// Register -> key map of the 32x32 C/D layout: lane (i = lane&31, h = lane>>5) holds S^T[j][i] for
// j = (r&3) + 8*(r>>2) + 4*h.  PV MFMA (jb, q) takes registers r = 8q..8q+7 as its 8 k-slots, so k-slot (h, e) is
// key 32*jb + ((8q+e)&3) + 8*((8q+e)>>2) + 4*h; the V^T image stores key j at column pos(j) = j with bits 2 and 3
// swapped, which makes those 8 keys one contiguous 16-byte read.
#include "cs_common.h"
#include "cs_f16x3.h"
#include <cstdlib>

namespace {

// image kernel: fragment-read ring depth (A/B builds: -DCS_ATTN_RING=0 leaves the schedule to the compiler)
#ifndef CS_ATTN_RING
#define CS_ATTN_RING 3
#endif
constexpr int RING = CS_ATTN_RING > 0 ? CS_ATTN_RING : 1;

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

constexpr float QK_SCALE = 16.0f;     // operand pre-scale (power of two) for Q*scale, K and V
constexpr float P_SCALE = 1024.0f;    // probabilities are <= 1

__device__ __forceinline__ void split1(float v, _Float16& hi, _Float16& lo) {
  hi = (_Float16)v;
  lo = (_Float16)(v - (float)hi);
}

// operand split that also tracks the lane's largest |operand| (overflow report, see CS_STATUS_F16X3_OVERFLOW)
__device__ __forceinline__ void split1m(float v, _Float16& hi, _Float16& lo, float& amax) {
  amax = fmaxf(amax, fabsf(v));
  split1(v, hi, lo);
}

__device__ __forceinline__ int vpos(int j) {   // swap bits 2 and 3
  return (j & ~12) | ((j & 4) << 1) | ((j & 8) >> 1);
}

// X1 = plain fp16 operands (hi halves only: one MFMA per product instead of three) -- the reduced-precision
// "fp16 MFMA attention" option of BASELINE configs[4]; not the default, outside the fp32 parity gates.
template <int DB, int KT, bool X1, int NW>
__global__ __launch_bounds__(64 * NW) void attn_f16x3_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                         const float* __restrict__ v, float* __restrict__ out,
                                                         int nq, int nk, int heads, int dh, int ldq, int ldk,
                                                         int ldv, int ldo, float scale, int qtiles, int nbh,
                                                         int32_t* __restrict__ status, float qs, float ks, float vs) {
  // qs / ks / vs (r5): the power-of-two operand pre-scales of Q * scale, K and V -- 16 each by default (QK_SCALE), or the
  // static bounds' scales of cs_attn_selfattn_f16x3_scaled (a transformer block's q / k / v are Linear(LayerNorm(x)):
  // bounded by the weights alone)
  constexpr int DP = 32 * DB;
  float amax = 0.f;                    // largest |scaled Q / K / V operand| this lane converted to fp16
  constexpr int LDK = DP + 8;          // halves; 16 consecutive rows hit 16 distinct 16-byte slots
  constexpr int LDV = KT + 8;
  constexpr int JB = KT / 32;
  constexpr int KS = DP / 16;          // k-steps of the QK^T contraction
  extern __shared__ __attribute__((aligned(16))) _Float16 sm[];
  _Float16* Kh = sm;                   // [KT][LDK]
  _Float16* Kl = Kh + KT * LDK;
  _Float16* Vh = Kl + KT * LDK;        // [DP][LDV]  (V^T, permuted key order)
  _Float16* Vl = Vh + DP * LDV;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int l31 = lane & 31;
  const int half = lane >> 5;

  // XCD-aware placement (speed only; see attn_f16x3_img_kernel): the query tiles of one (sample, head) read the same K / V
  // rows, so they go to ONE XCD (block w runs on XCD w % 8) and fetch them through the fabric once, not once per tile
#ifdef CS_ATTN_NO_XCD
  int bid = blockIdx.x;
  const int qt = bid % qtiles;
  bid /= qtiles;
#else
  const int xw = blockIdx.x & 7, xj = blockIdx.x >> 3;
  const int qt = xj % qtiles;
  const int bid = xw + 8 * (xj / qtiles);
  if (bid >= nbh) return;                              // grid padded to a multiple of 8 groups (whole workgroup exits)
#endif
  const int h = bid % heads;
  const int b = bid / heads;

  constexpr int NT = 64 * NW;          // NW waves of 32 queries share each staged K / V tile
  const int q0 = qt * (32 * NW) + wave * 32;
  const int qi = min(q0 + l31, nq - 1);
  const float* qp = q + ((int64_t)b * nq + qi) * ldq + h * dh;
  const float* kb = k + (int64_t)b * nk * ldk + h * dh;
  const float* vb = v + (int64_t)b * nk * ldv + h * dh;

  // Q fragments: qh[t][e] = Q[i][16t + 8*half + e] * scale * QK_SCALE
  h8 qh[KS], ql[KS];
#pragma unroll
  for (int t = 0; t < KS; ++t)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int d = 16 * t + 8 * half + e;
      const float x = d < dh ? qp[d] * (scale * qs) : 0.f;
      _Float16 a, c;
      split1m(x, a, c, amax);
      qh[t][e] = a;
      ql[t][e] = c;
    }

  // zero the padding (d >= dh) of the K images and the V^T images once
  for (int u = tid; u < KT * (LDK - dh); u += NT) {
    const int j = u / (LDK - dh);
    const int d = dh + (u - j * (LDK - dh));
    Kh[j * LDK + d] = (_Float16)0.f;
    Kl[j * LDK + d] = (_Float16)0.f;
  }
  for (int u = tid; u < (DP - dh) * LDV; u += NT) {
    Vh[dh * LDV + u] = (_Float16)0.f;
    Vl[dh * LDV + u] = (_Float16)0.f;
  }

  f32x16 oacc[DB];
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
  // Online softmax in the accumulator's own units: sacc = logit * QK_SCALE^2, so with c = log2(e) / QK_SCALE^2
  //   p * P_SCALE = exp2((sacc - m) * c + log2(P_SCALE))
  // is one subtract, one fma and one v_exp_f32 per element (the row maximum m stays in raw units, the P_SCALE the
  // PV operands need is folded into the exponent, and the running sum carries it until the final normalisation).
  float mrun = -INFINITY;
  float lrun = 0.f;
  const float cexp = 1.44269504088896340736f / (qs * ks);
  const float lp = 10.0f;             // log2(P_SCALE)
  static_assert(P_SCALE == 1024.0f, "lp = log2(P_SCALE)");

  // K / V tile staging, software-pipelined: the global loads of tile t+1 are issued (into registers) before tile t's
  // MFMA / softmax work and only converted + written to LDS after it, so their latency never sits between barriers.
  // Unit i of a thread: K -> (key j, 4 channels), coalesced along d; V -> (4 channels, key j) with the key fastest,
  // so a wave writes one contiguous run of a V^T row.
  const int dh4 = dh >> 2;
  constexpr bool PIPE = DB <= 2;
  constexpr int NU = PIPE ? (KT * DP / 4 + NT - 1) / NT : 1;   // units per thread per tensor (KT * DP/4 units over NT threads)
  int k_g[NU], k_l[NU], v_g[NU], v_l[NU];
#pragma unroll
  for (int i = 0; i < NU; ++i) {
    const int u = tid + NT * i;
    const bool ok = u < KT * dh4;
    const int j = ok ? u / dh4 : 0;
    const int c4 = ok ? u - j * dh4 : 0;
    k_g[i] = j * ldk + c4 * 4;
    k_l[i] = ok ? j * LDK + c4 * 4 : -1;
    const int vc4 = ok ? u / KT : 0;
    const int vj = ok ? u - vc4 * KT : 0;
    v_g[i] = vj * ldv + vc4 * 4;
    v_l[i] = ok ? (vc4 * 4) * LDV + vpos(vj) : -1;
  }
  float4 kr[NU], vr[NU];
  auto load_tile = [&](int kt0) {
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      kr[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      vr[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      // key index of the unit, recovered from its LDS offset (only matters on the ragged last tile)
      if (k_l[i] >= 0 && kt0 + k_l[i] / LDK < nk)
        kr[i] = *reinterpret_cast<const float4*>(kb + (int64_t)kt0 * ldk + k_g[i]);
      if (v_l[i] >= 0 && kt0 + vpos(v_l[i] % LDV) < nk)
        vr[i] = *reinterpret_cast<const float4*>(vb + (int64_t)kt0 * ldv + v_g[i]);
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      if (k_l[i] >= 0) {
        h4 hi, lo;
        _Float16 a, c;
        split1m(kr[i].x * ks, a, c, amax); hi[0] = a; lo[0] = c;
        split1m(kr[i].y * ks, a, c, amax); hi[1] = a; lo[1] = c;
        split1m(kr[i].z * ks, a, c, amax); hi[2] = a; lo[2] = c;
        split1m(kr[i].w * ks, a, c, amax); hi[3] = a; lo[3] = c;
        *reinterpret_cast<h4*>(Kh + k_l[i]) = hi;
        if constexpr (!X1) *reinterpret_cast<h4*>(Kl + k_l[i]) = lo;
      }
      if (v_l[i] >= 0) {
        const float x[4] = {vr[i].x, vr[i].y, vr[i].z, vr[i].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          _Float16 a, c;
          split1m(x[e] * vs, a, c, amax);
          Vh[v_l[i] + e * LDV] = a;
          if constexpr (!X1) Vl[v_l[i] + e * LDV] = c;
        }
      }
    }
  };
  // The prefetch costs 12 * NU registers: worth it while the kernel keeps >= 2 waves per SIMD (DB <= 2: 852 -> 661 us
  // at 1024 tokens x dh 56), a loss once it drops the wider variants to one (dh 84: 105 -> 119 us), so those load
  // and convert in place.
  if (PIPE) load_tile(0);
  for (int kt0 = 0; kt0 < nk; kt0 += KT) {
    __syncthreads();  // previous tile fully consumed
    if constexpr (PIPE) {
#ifdef CS_ATTN_WHATIF_NO_STAGE         // (timing-only what-if builds, tools/attn_unet_whatif.sh: wrong results)
      if (kt0 == 0)
#endif
      store_tile();
    } else {
      for (int u = tid; u < KT * dh4; u += NT) {          // K: unit = (key j, 4 channels)
        const int j = u / dh4;
        const int c4 = u - j * dh4;
        float4 kv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (kt0 + j < nk) kv = *reinterpret_cast<const float4*>(kb + (int64_t)(kt0 + j) * ldk + c4 * 4);
        h4 hi, lo;
        _Float16 a, c;
        split1m(kv.x * ks, a, c, amax); hi[0] = a; lo[0] = c;
        split1m(kv.y * ks, a, c, amax); hi[1] = a; lo[1] = c;
        split1m(kv.z * ks, a, c, amax); hi[2] = a; lo[2] = c;
        split1m(kv.w * ks, a, c, amax); hi[3] = a; lo[3] = c;
        *reinterpret_cast<h4*>(Kh + j * LDK + c4 * 4) = hi;
        if constexpr (!X1) *reinterpret_cast<h4*>(Kl + j * LDK + c4 * 4) = lo;
      }
      for (int u = tid; u < KT * dh4; u += NT) {          // V: unit = (4 channels, key j), key fastest
        const int c4 = u / KT;
        const int j = u - c4 * KT;
        float4 vv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (kt0 + j < nk) vv = *reinterpret_cast<const float4*>(vb + (int64_t)(kt0 + j) * ldv + c4 * 4);
        const int pj = vpos(j);
        const float x[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          _Float16 a, c;
          split1m(x[i] * vs, a, c, amax);
          Vh[(c4 * 4 + i) * LDV + pj] = a;
          if constexpr (!X1) Vl[(c4 * 4 + i) * LDV + pj] = c;
        }
      }
    }
    __syncthreads();
#ifdef CS_ATTN_WHATIF_NO_STAGE
    if (PIPE && kt0 == 0) load_tile(KT);
#else
    if (PIPE && kt0 + KT < nk) load_tile(kt0 + KT);
#endif

    // ---- S^T = K Q^T ----
    f32x16 sacc[JB];
#pragma unroll
    for (int jb = 0; jb < JB; ++jb)
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[jb][r] = 0.f;
#pragma unroll
    for (int t = 0; t < KS; ++t)
#pragma unroll
      for (int jb = 0; jb < JB; ++jb) {
        const int off = (jb * 32 + l31) * LDK + 16 * t + 8 * half;
        const h8 kh = *reinterpret_cast<const h8*>(Kh + off);
        if constexpr (!X1) {
          const h8 kl = *reinterpret_cast<const h8*>(Kl + off);
          sacc[jb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[t], sacc[jb], 0, 0, 0);
          sacc[jb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[t], sacc[jb], 0, 0, 0);
        }
        sacc[jb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[t], sacc[jb], 0, 0, 0);
      }

    // ---- online softmax (per query i = lane&31) ----
    if (kt0 + KT > nk) {               // ragged last tile only (wave-uniform): keys past nk take no weight
#pragma unroll
      for (int jb = 0; jb < JB; ++jb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int j = kt0 + jb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          if (j >= nk) sacc[jb][r] = -INFINITY;
        }
    }
    float mloc = -INFINITY;
#pragma unroll
    for (int jb = 0; jb < JB; ++jb)
#pragma unroll
      for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, sacc[jb][r]);
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
    const float mnew = fmaxf(mrun, mloc);
    const float alpha = (mrun == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f((mrun - mnew) * cexp);
    float psum = 0.f;
#pragma unroll
    for (int jb = 0; jb < JB; ++jb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
#ifdef CS_ATTN_WHATIF_NO_EXP
        const float pv = fmaf(sacc[jb][r] - mnew, cexp, lp);
#else
        const float pv = __builtin_amdgcn_exp2f(fmaf(sacc[jb][r] - mnew, cexp, lp));   // = p * P_SCALE
#endif
        sacc[jb][r] = pv;
        psum += pv;
      }
    lrun = lrun * alpha + psum;
    mrun = mnew;
#ifndef CS_ATTN_WHATIF_NO_RESCALE
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[d][r] *= alpha;
#endif

    // ---- O^T += V^T P^T ----
#pragma unroll
    for (int jb = 0; jb < JB; ++jb)
#pragma unroll
      for (int qq = 0; qq < 2; ++qq) {
        h8 ph, pl;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          _Float16 a, c;
          split1(sacc[jb][8 * qq + e], a, c);
          ph[e] = a;
          pl[e] = c;
        }
#pragma unroll
        for (int d = 0; d < DB; ++d) {
          const int off = (32 * d + l31) * LDV + 32 * jb + 16 * qq + 8 * half;
          const h8 vh = *reinterpret_cast<const h8*>(Vh + off);
          if constexpr (!X1) {
            const h8 vl = *reinterpret_cast<const h8*>(Vl + off);
            oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph, oacc[d], 0, 0, 0);
            oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl, oacc[d], 0, 0, 0);
          }
          oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph, oacc[d], 0, 0, 0);
        }
      }
  }

  if (status && amax >= 65504.f) atomicOr(status, CS_STATUS_F16X3_OVERFLOW);
  const float ltot = lrun + __shfl_xor(lrun, 32, 64);
  const float inv = 1.0f / (ltot * vs);       // ltot already carries P_SCALE
  if (q0 + l31 < nq) {
    float* op = out + ((int64_t)b * nq + q0 + l31) * ldo + h * dh;
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int dd = 32 * d + 8 * g + 4 * half;
        if (dd < dh) {
          float4 o;
          o.x = oacc[d][4 * g + 0] * inv;
          o.y = oacc[d][4 * g + 1] * inv;
          o.z = oacc[d][4 * g + 2] * inv;
          o.w = oacc[d][4 * g + 3] * inv;
          *reinterpret_cast<float4*>(op + dd) = o;
        }
      }
  }
}

// ---------------------------------------------------------------------------------------------------------
// r3: K / V split ONCE per call.  The kernel above converts every K / V tile to its fp16 hi / lo LDS images inside
// every workgroup -- qtiles-fold redundant VALU work and (V^T) two-byte scattered LDS stores: for the VQ decoder's
// single 256-channel head over 4096 tokens that staging was ~7/8 of the kernel (95 TF/s).  Here a pre-pass writes,
// per (sample, head, key tile), the EXACT LDS image the MFMA loop reads ([Kh | Kl | Vh | Vl], padding included), and
// the attention kernel pulls whole images into a two-deep LDS ring with buffer_load ... lds (no registers, no VALU),
// the next tile in flight under the current tile's MFMAs.  Same operand values, same MFMA order: bit-identical
// output (tests/test_f16x3_gpu.py).
// ---------------------------------------------------------------------------------------------------------
template <int DB, int KT>
struct AttnImg {
  static constexpr int DP = 32 * DB, LDK = DP + 8, LDV = KT + 8;
  static constexpr int TILE_HALVES = 2 * (KT * LDK + DP * LDV);
  static constexpr int TILE_BYTES = TILE_HALVES * 2;
  static_assert(TILE_BYTES % 1024 == 0, "whole 1 KB DMA chunks");
};

template <int DB, int KT>
__global__ __launch_bounds__(256) void attn_presplit_kernel(const float* __restrict__ k, const float* __restrict__ v,
                                                            _Float16* __restrict__ img, int nk, int heads, int dh,
                                                            int ldk, int ldv, int ntiles, int32_t* __restrict__ status,
                                                            float ks, float vs) {
  // ks / vs (r6): the operand pre-scales of K and V (QK_SCALE = 16 by default; static bounds: cs_attn_selfattn_f16x3_ws_scaled)
  using I = AttnImg<DB, KT>;
  constexpr int LDK = I::LDK, LDV = I::LDV, DP = I::DP;
  extern __shared__ __attribute__((aligned(16))) _Float16 sm[];
  _Float16* Kh = sm;
  _Float16* Kl = Kh + KT * LDK;
  _Float16* Vh = Kl + KT * LDK;
  _Float16* Vl = Vh + DP * LDV;
  const int tid = threadIdx.x;
  int bid = blockIdx.x;
  const int tile = bid % ntiles;
  bid /= ntiles;
  const int h = bid % heads;
  const int b = bid / heads;
  const int kt0 = tile * KT;
  const float* kb = k + (int64_t)b * nk * ldk + h * dh;
  const float* vb = v + (int64_t)b * nk * ldv + h * dh;
  float amax = 0.f;
  for (int u = tid; u < I::TILE_BYTES / 16; u += 256) reinterpret_cast<uint4*>(sm)[u] = make_uint4(0u, 0u, 0u, 0u);
  __syncthreads();
  const int dh4 = dh >> 2;
  for (int u = tid; u < KT * dh4; u += 256) {          // K: unit = (key j, 4 channels)
    const int j = u / dh4;
    const int c4 = u - j * dh4;
    float4 kv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (kt0 + j < nk) kv = *reinterpret_cast<const float4*>(kb + (int64_t)(kt0 + j) * ldk + c4 * 4);
    h4 hi, lo;
    _Float16 a, c;
    split1m(kv.x * ks, a, c, amax); hi[0] = a; lo[0] = c;
    split1m(kv.y * ks, a, c, amax); hi[1] = a; lo[1] = c;
    split1m(kv.z * ks, a, c, amax); hi[2] = a; lo[2] = c;
    split1m(kv.w * ks, a, c, amax); hi[3] = a; lo[3] = c;
    *reinterpret_cast<h4*>(Kh + j * LDK + c4 * 4) = hi;
    *reinterpret_cast<h4*>(Kl + j * LDK + c4 * 4) = lo;
  }
  for (int u = tid; u < KT * dh4; u += 256) {          // V: unit = (key j, 4 channels) too -- coalesced reads
    const int j = u / dh4;
    const int c4 = u - j * dh4;
    float4 vv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (kt0 + j < nk) vv = *reinterpret_cast<const float4*>(vb + (int64_t)(kt0 + j) * ldv + c4 * 4);
    const int pj = vpos(j);
    const float x[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      _Float16 a, c;
      split1m(x[i] * vs, a, c, amax);
      Vh[(c4 * 4 + i) * LDV + pj] = a;
      Vl[(c4 * 4 + i) * LDV + pj] = c;
    }
  }
  __syncthreads();
  uint4* dst = reinterpret_cast<uint4*>(img + (int64_t)blockIdx.x * I::TILE_HALVES);
  for (int u = tid; u < I::TILE_BYTES / 16; u += 256) dst[u] = reinterpret_cast<const uint4*>(sm)[u];
  if (status && amax >= 65504.f) atomicOr(status, CS_STATUS_F16X3_OVERFLOW);
}

template <int DB, int KT, int NW>
__global__ __launch_bounds__(64 * NW) void attn_f16x3_img_kernel(const float* __restrict__ q,
                                                                 const _Float16* __restrict__ img,
                                                                 float* __restrict__ out, int nq, int nk, int heads,
                                                                 int dh, int ldq, int ldo, float scale, int qtiles,
                                                                 int ntiles, int nbh, int32_t* __restrict__ status,
                                                                 float qs, float ks, float vs) {
  using I = AttnImg<DB, KT>;
  constexpr int DP = I::DP, LDK = I::LDK, LDV = I::LDV;
  constexpr int JB = KT / 32;
  constexpr int KS = DP / 16;
  // LDS: two K slots [Kh | Kl] then two V slots [Vh | Vl].  The K images run ONE TILE AHEAD of the V images: iteration t
  // computes S(t + 1) = K(t + 1) Q^T next to the softmax of S(t) (independent work for the one wave a SIMD holds: MFMA
  // under VALU) and then O += V(t) P(t); K(t + 2) and V(t + 1) are in flight meanwhile.  Same per-tile operations in the
  // same order as attn_f16x3_kernel: bit-identical output.
  constexpr int KBYTES = 4 * KT * LDK, VBYTES = 4 * DP * LDV;
  static_assert(KBYTES + VBYTES == I::TILE_BYTES && KBYTES % 1024 == 0 && VBYTES % 1024 == 0, "image layout");
  constexpr int KCH = KBYTES / 1024, VCH = VBYTES / 1024;      // 1 KB (one wave-wide 16-byte DMA) chunks
  extern __shared__ __attribute__((aligned(16))) unsigned char smb[];
  float amax = 0.f;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar: the DMA chunk loops branch on it
  const int l31 = lane & 31;
  const int half = lane >> 5;

  // XCD-aware placement (speed only): block w runs on XCD w % 8, each XCD has its own 4 MB L2, and the query tiles of
  // one (sample, head) all stream the SAME tile images -- so the j-th block of XCD x takes query tile j % qtiles of group
  // x + 8 * (j / qtiles): a group's workgroups share one L2 and run side by side (the decoder: 32 query tiles = the 32 CUs
  // of an XCD).  In launch order every XCD would hold tiles of eight groups at once (77 MB of images against 4 MB of
  // L2: each workgroup then pulls its 9.6 MB through the fabric, ~5 TB/s chip-wide -- that was the kernel's bound).
#ifdef CS_ATTN_NO_XCD
  int bid = blockIdx.x;
  const int qt = bid % qtiles;
  bid /= qtiles;
#else
  const int xw = blockIdx.x & 7, xj = blockIdx.x >> 3;
  const int qt = xj % qtiles;
  const int bid = xw + 8 * (xj / qtiles);
  if (bid >= nbh) return;                              // grid padded to a multiple of 8 groups (whole workgroup exits)
#endif
  const int h = bid % heads;
  const int b = bid / heads;

  const int q0 = qt * (32 * NW) + wave * 32;
  const int qi = min(q0 + l31, nq - 1);
  const float* qp = q + ((int64_t)b * nq + qi) * ldq + h * dh;

  // this (sample, head)'s tile images: one buffer window, 32-bit offsets (host checks ntiles * TILE_BYTES < 2^31)
  const __amdgpu_buffer_rsrc_t irs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(img + ((int64_t)b * heads + h) * ntiles * I::TILE_HALVES), 0, (unsigned)ntiles * (unsigned)I::TILE_BYTES,
      0x00020000);
  auto dma_k = [&](int t) {                            // K image of tile t -> K slot t & 1; chunk c by wave c % NW
    unsigned char* dst = smb + (t & 1) * KBYTES;
    const unsigned base = (unsigned)t * (unsigned)I::TILE_BYTES + (unsigned)lane * 16u;
#pragma unroll
    for (int i = 0; i < (KCH + NW - 1) / NW; ++i) {
      const int c = wave + NW * i;                     // wave-uniform
      if (c < KCH) __builtin_amdgcn_raw_ptr_buffer_load_lds(irs, dst + c * 1024, 16, base + (unsigned)c * 1024u, 0, 0, 0);
    }
  };
  auto dma_v = [&](int t) {                            // V image of tile t -> V slot t & 1
    unsigned char* dst = smb + 2 * KBYTES + (t & 1) * VBYTES;
    const unsigned base = (unsigned)t * (unsigned)I::TILE_BYTES + (unsigned)KBYTES + (unsigned)lane * 16u;
#pragma unroll
    for (int i = 0; i < (VCH + NW - 1) / NW; ++i) {
      const int c = wave + NW * i;
      if (c < VCH) __builtin_amdgcn_raw_ptr_buffer_load_lds(irs, dst + c * 1024, 16, base + (unsigned)c * 1024u, 0, 0, 0);
    }
  };
  dma_k(0);
  dma_v(0);

  h8 qh[KS], ql[KS];
#pragma unroll
  for (int t = 0; t < KS; ++t)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int d = 16 * t + 8 * half + e;
      const float x = d < dh ? qp[d] * (scale * qs) : 0.f;
      _Float16 a, c;
      split1m(x, a, c, amax);
      qh[t][e] = a;
      ql[t][e] = c;
    }

  f32x16 oacc[DB];
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
  float mrun = -INFINITY;
  float lrun = 0.f;
  const float cexp = 1.44269504088896340736f / (qs * ks);
  const float lp = 10.0f;             // log2(P_SCALE)

  // S^T = K Q^T of tile t (its K image sits in K slot t & 1)
  auto qk = [&](int t, f32x16 (&sacc)[JB]) {
    const _Float16* Kh = reinterpret_cast<const _Float16*>(smb + (t & 1) * KBYTES);
    const _Float16* Kl = Kh + KT * LDK;
#pragma unroll
    for (int jb = 0; jb < JB; ++jb)
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[jb][r] = 0.f;
#if CS_ATTN_RING
    // fragment reads RING steps ahead of their MFMAs; the scheduling fences keep the compiler from sinking every read
    // next to its use (at 512 live registers it minimises live ranges: read -> wait -> MFMA, the whole LDS latency exposed)
    constexpr int NS = KS * JB;
    h8 ah[RING], al[RING];
    auto ld = [&](int st, int slot) {
      const int tt = st / JB, jb = st - tt * JB;
      const int off = (jb * 32 + l31) * LDK + 16 * tt + 8 * half;
#ifdef CS_ATTN_WHATIF_NO_LDSREAD
      ah[slot] = qh[tt]; al[slot] = ql[tt]; (void)off;
#else
      ah[slot] = *reinterpret_cast<const h8*>(Kh + off);
      al[slot] = *reinterpret_cast<const h8*>(Kl + off);
#endif
    };
#pragma unroll
    for (int st = 0; st < RING && st < NS; ++st) ld(st, st);
#pragma unroll
    for (int st = 0; st < NS; ++st) {
      const int tt = st / JB, jb = st - tt * JB, slot = st % RING;
      __builtin_amdgcn_sched_barrier(0);
      const h8 kh = ah[slot], kl = al[slot];
      sacc[jb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[tt], sacc[jb], 0, 0, 0);
      sacc[jb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[tt], sacc[jb], 0, 0, 0);
      sacc[jb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[tt], sacc[jb], 0, 0, 0);
      if (st + RING < NS) ld(st + RING, slot);
    }
    __builtin_amdgcn_sched_barrier(0);
#else
#pragma unroll
    for (int tt = 0; tt < KS; ++tt)
#pragma unroll
      for (int jb = 0; jb < JB; ++jb) {
        const int off = (jb * 32 + l31) * LDK + 16 * tt + 8 * half;
        const h8 kh = *reinterpret_cast<const h8*>(Kh + off);
        const h8 kl = *reinterpret_cast<const h8*>(Kl + off);
        sacc[jb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[tt], sacc[jb], 0, 0, 0);
        sacc[jb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[tt], sacc[jb], 0, 0, 0);
        sacc[jb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[tt], sacc[jb], 0, 0, 0);
      }
#endif
  };

  f32x16 scur[JB], snext[JB];
  cs16::wait_vmcnt<0>();
  __syncthreads();                    // K(0), V(0) have landed for every wave
  if (ntiles > 1) dma_k(1);
  qk(0, scur);

  for (int t = 0; t < ntiles; ++t) {
    const int kt0 = t * KT;
    cs16::wait_vmcnt<0>();            // this wave's chunks of K(t + 1) (and V(t), t > 0) have landed ...
#ifndef CS_ATTN_WHATIF_NO_BARRIER
    __syncthreads();                  // ... everyone's have, and everyone is done with K(t) and V(t - 1)
#endif
#ifndef CS_ATTN_WHATIF_NO_DMA          // (timing-only what-if builds below: wrong results)
    if (t + 2 < ntiles) dma_k(t + 2);
    if (t + 1 < ntiles) dma_v(t + 1);
#endif
    if (t + 1 < ntiles) qk(t + 1, snext);

    // ---- online softmax of tile t (per query i = lane&31) ----
    if (kt0 + KT > nk) {               // ragged last tile only (wave-uniform): keys past nk take no weight
#pragma unroll
      for (int jb = 0; jb < JB; ++jb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int j = kt0 + jb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          if (j >= nk) scur[jb][r] = -INFINITY;
        }
    }
    float mloc = -INFINITY;
#pragma unroll
    for (int jb = 0; jb < JB; ++jb)
#pragma unroll
      for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, scur[jb][r]);
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
    const float mnew = fmaxf(mrun, mloc);
    const float alpha = (mrun == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f((mrun - mnew) * cexp);
    float psum = 0.f;
#pragma unroll
    for (int jb = 0; jb < JB; ++jb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
#ifdef CS_ATTN_WHATIF_NO_EXP           // (timing-only what-if build: wrong results)
        const float pv = fmaf(scur[jb][r] - mnew, cexp, lp);
#else
        const float pv = __builtin_amdgcn_exp2f(fmaf(scur[jb][r] - mnew, cexp, lp));   // = p * P_SCALE
#endif
        scur[jb][r] = pv;
        psum += pv;
      }
    lrun = lrun * alpha + psum;
    mrun = mnew;
#ifndef CS_ATTN_WHATIF_NO_RESCALE      // (timing-only what-if build: wrong results)
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[d][r] *= alpha;
#endif

    // ---- O^T += V(t)^T P^T ----
    const _Float16* Vh = reinterpret_cast<const _Float16*>(smb + 2 * KBYTES + (t & 1) * VBYTES);
    const _Float16* Vl = Vh + DP * LDV;
#if CS_ATTN_RING
    {
      constexpr int NS = JB * 2 * DB;                    // step = ((jb, qq), d)
      h8 ph[JB * 2], pl[JB * 2];
#pragma unroll
      for (int g = 0; g < JB * 2; ++g)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          _Float16 a, c;
          split1(scur[g >> 1][8 * (g & 1) + e], a, c);
          ph[g][e] = a;
          pl[g][e] = c;
        }
      h8 ah[RING], al[RING];
      auto ld = [&](int st, int slot) {
        const int g = st / DB, d = st - g * DB;
        const int off = (32 * d + l31) * LDV + 32 * (g >> 1) + 16 * (g & 1) + 8 * half;
#ifdef CS_ATTN_WHATIF_NO_LDSREAD
        ah[slot] = qh[d]; al[slot] = ql[d]; (void)off;
#else
        ah[slot] = *reinterpret_cast<const h8*>(Vh + off);
        al[slot] = *reinterpret_cast<const h8*>(Vl + off);
#endif
      };
#pragma unroll
      for (int st = 0; st < RING && st < NS; ++st) ld(st, st);
#pragma unroll
      for (int st = 0; st < NS; ++st) {
        const int g = st / DB, d = st - g * DB, slot = st % RING;
        __builtin_amdgcn_sched_barrier(0);
        const h8 vh = ah[slot], vl = al[slot];
        oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph[g], oacc[d], 0, 0, 0);
        oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl[g], oacc[d], 0, 0, 0);
        oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph[g], oacc[d], 0, 0, 0);
        if (st + RING < NS) ld(st + RING, slot);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#else
#pragma unroll
    for (int jb = 0; jb < JB; ++jb)
#pragma unroll
      for (int qq = 0; qq < 2; ++qq) {
        h8 ph, pl;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          _Float16 a, c;
          split1(scur[jb][8 * qq + e], a, c);
          ph[e] = a;
          pl[e] = c;
        }
#pragma unroll
        for (int d = 0; d < DB; ++d) {
          const int off = (32 * d + l31) * LDV + 32 * jb + 16 * qq + 8 * half;
          const h8 vh = *reinterpret_cast<const h8*>(Vh + off);
          const h8 vl = *reinterpret_cast<const h8*>(Vl + off);
          oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph, oacc[d], 0, 0, 0);
          oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl, oacc[d], 0, 0, 0);
          oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph, oacc[d], 0, 0, 0);
        }
      }
#endif
#pragma unroll
    for (int jb = 0; jb < JB; ++jb) scur[jb] = snext[jb];
  }

  if (status && amax >= 65504.f) atomicOr(status, CS_STATUS_F16X3_OVERFLOW);
  const float ltot = lrun + __shfl_xor(lrun, 32, 64);
  const float inv = 1.0f / (ltot * vs);       // ltot already carries P_SCALE
  if (q0 + l31 < nq) {
    float* op = out + ((int64_t)b * nq + q0 + l31) * ldo + h * dh;
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int dd = 32 * d + 8 * g + 4 * half;
        if (dd < dh) {
          float4 o;
          o.x = oacc[d][4 * g + 0] * inv;
          o.y = oacc[d][4 * g + 1] * inv;
          o.z = oacc[d][4 * g + 2] * inv;
          o.w = oacc[d][4 * g + 3] * inv;
          *reinterpret_cast<float4*>(op + dd) = o;
        }
      }
  }
}

template <int DB, int KT, int NW>
int launch_attn16_img(const float* q, const float* k, const float* v, float* out, int nb, int nq, int nk, int heads,
                      int dh, int ldq, int ldk, int ldv, int ldo, float scale, int32_t* status, void* ws, hipStream_t s,
                      float qs = QK_SCALE, float ks = QK_SCALE, float vs = QK_SCALE) {
  using I = AttnImg<DB, KT>;
  const int ntiles = (nk + KT - 1) / KT;
  if ((int64_t)ntiles * I::TILE_BYTES >= 0x7FF00000LL) return CS_EINVAL;
  const int64_t pgrid = (int64_t)nb * heads * ntiles;
  const int qtiles = (nq + 32 * NW - 1) / (32 * NW);
#ifdef CS_ATTN_NO_XCD
  const int64_t grid = (int64_t)qtiles * heads * nb;
#else
  const int64_t grid = (int64_t)qtiles * (((int64_t)heads * nb + 7) / 8 * 8);      // whole groups of eight (sample, head)s
#endif
  if (pgrid > 0x7fffffffLL || grid > 0x7fffffffLL) return CS_EINVAL;
  auto pk = attn_presplit_kernel<DB, KT>;
  auto kern = attn_f16x3_img_kernel<DB, KT, NW>;
  {
    // raise the dynamic-LDS limits on every call, as launch_attn16 does: the attribute is per device, a process may drive
    // several (the status word is keyed per device), and a `static bool once` guard is a data race under concurrent
    // callers.  The call is a host-side table update (~1 us), nothing next to the two launches below.
    hipError_t e = hipFuncSetAttribute((const void*)pk, hipFuncAttributeMaxDynamicSharedMemorySize, I::TILE_BYTES);
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * I::TILE_BYTES);
    if (e != hipSuccess) return (int)e;
  }
  CS_LAUNCH(pk, dim3((unsigned)pgrid), dim3(256), (size_t)I::TILE_BYTES, s, k, v, (_Float16*)ws, nk, heads, dh, ldk, ldv,
            ntiles, status, ks, vs);
  CS_CHECK_LAUNCH();
  CS_LAUNCH(kern, dim3((unsigned)grid), dim3(64 * NW), (size_t)2 * I::TILE_BYTES, s, q, (const _Float16*)ws, out, nq, nk,
            heads, dh, ldq, ldo, scale, qtiles, ntiles, nb * heads, status, qs, ks, vs);
  CS_CHECK_LAUNCH();
  return CS_OK;
}

// which (DB, KT, NW) the image path runs for a head width / query count, 0 = none (the in-kernel split above)
inline int img_variant(int nq, int nk, int dh) {
  if (cs_debug()->no_attn_img) return 0;
  // Only where the pre-pass is amortised over many query tiles AND the in-kernel split dominates: the 256-wide variant
  // (one wave per SIMD, nothing overlaps its staging) from 8 query tiles up -- the VQ decoder's mid attention, 4096 tokens:
  // 2385 -> 1005 us per 16 objects (115 -> 270 TF/s).  Measured and left on the in-kernel split: dh 56 at 1024 tokens
  // (4 query tiles per head: 640 -> 668 us, the pre-pass costs more than the staging it removes).
  if (dh > 128 && dh <= 256 && nq >= 1024 && nk >= 128) return 8;
  return 0;
}

template <int DB, int KT, bool X1, int NW = 4>
int launch_attn16(const float* q, const float* k, const float* v, float* out, int nb, int nq, int nk, int heads,
                  int dh, int ldq, int ldk, int ldv, int ldo, float scale, int32_t* status, hipStream_t s,
                  float qs = QK_SCALE, float ks = QK_SCALE, float vs = QK_SCALE) {
  constexpr int DP = 32 * DB;
  const size_t smem = (size_t)(2 * KT * (DP + 8) + 2 * DP * (KT + 8)) * sizeof(_Float16);
  const int qtiles = (nq + 32 * NW - 1) / (32 * NW);
#ifdef CS_ATTN_NO_XCD
  const int64_t grid = (int64_t)qtiles * heads * nb;
#else
  const int64_t grid = (int64_t)qtiles * (((int64_t)heads * nb + 7) / 8 * 8);      // whole groups of eight (sample, head)s
#endif
  if (grid > 0x7fffffffLL) return CS_EINVAL;
  auto kern = attn_f16x3_kernel<DB, KT, X1, NW>;
  if (smem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
  }
  CS_LAUNCH(kern, dim3((unsigned)grid), dim3(64 * NW), smem, s, q, k, v, out, nq, nk, heads, dh, ldq, ldk, ldv, ldo,
            scale, qtiles, nb * heads, status, qs, ks, vs);
  CS_CHECK_LAUNCH();
  return CS_OK;
}

}  // namespace

template <bool X1>
static int attn16_dispatch(const float* q, const float* k, const float* v, float* out, int nb, int nq, int nk, int heads,
                           int dh, int ldq, int ldk, int ldv, int ldo, float scale, int32_t* status, cs_stream_t stream,
                           float qs = QK_SCALE, float ks = QK_SCALE, float vs = QK_SCALE) {
  if (!q || !k || !v || !out || nb <= 0 || nq <= 0 || nk <= 0 || heads <= 0 || dh <= 0) return CS_EINVAL;
  if ((dh & 3) || (ldq & 3) || (ldk & 3) || (ldv & 3) || (ldo & 3)) return CS_EINVAL;
  if (ldq < heads * dh || ldk < heads * dh || ldv < heads * dh || ldo < heads * dh) return CS_EINVAL;
  if (((uintptr_t)q & 15) || ((uintptr_t)k & 15) || ((uintptr_t)v & 15) || ((uintptr_t)out & 15))
    return CS_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (dh <= 32) return launch_attn16<1, 64, X1>(q, k, v, out, nb, nq, nk, heads, dh, ldq, ldk, ldv, ldo, scale, status, s, qs, ks, vs);
  // eight waves per staged tile amortise the K / V conversion over 256 queries -- but only once such workgroups fill the
  // chip: at small batches (1 object: 16 (sample, head) groups, 64 workgroups at 1024 tokens, 16 at 256) four-wave
  // workgroups double the count on a mostly idle GPU (1 object: 71.4 -> 60.0 us and 34.5 -> 31.7; from 128 eight-wave
  // workgroups up they lose: 224 -> 448 workgroups at 14 objects' 256-token level 48.7 -> 55.2 us;
  // profiles/r03_ar_attn_nw.txt).  Same per-query arithmetic either way: bit-identical.
  // CS_ATTN_NW8=1: the previous rule (eight waves from 512 / 256 queries), A/B runs.
  const bool fill8 = cs_debug()->attn_nw8 || (int64_t)nb * heads * ((nq + 255) / 256) >= 128;
  // (r5, measured and not taken: TWO-wave workgroups at one or two objects -- every CU gets one -- ran 62.3 vs 55.4 us at 1024
  // tokens x dh 56 and 49.7 vs 38.2 us at 256 x dh 84: the K / V staging per workgroup does not shrink with the query tile,
  // profiles/r05_f_attn_nw2_ab.txt)
  if (dh <= 64) {
    if (nq >= 512 && fill8)
      return launch_attn16<2, 64, X1, 8>(q, k, v, out, nb, nq, nk, heads, dh, ldq, ldk, ldv, ldo, scale, status, s, qs, ks, vs);
    return launch_attn16<2, 64, X1>(q, k, v, out, nb, nq, nk, heads, dh, ldq, ldk, ldv, ldo, scale, status, s, qs, ks, vs);
  }
  if (dh <= 96) {
    if (nq >= 256 && fill8)
      return launch_attn16<3, 64, X1, 8>(q, k, v, out, nb, nq, nk, heads, dh, ldq, ldk, ldv, ldo, scale, status, s, qs, ks, vs);
    return launch_attn16<3, 64, X1>(q, k, v, out, nb, nq, nk, heads, dh, ldq, ldk, ldv, ldo, scale, status, s, qs, ks, vs);
  }
  if (dh <= 128) return launch_attn16<4, 64, X1>(q, k, v, out, nb, nq, nk, heads, dh, ldq, ldk, ldv, ldo, scale, status, s, qs, ks, vs);
  if (dh <= 256) return launch_attn16<8, 32, X1>(q, k, v, out, nb, nq, nk, heads, dh, ldq, ldk, ldv, ldo, scale, status, s, qs, ks, vs);
  return CS_EINVAL;
}

extern "C" int cs_attn_selfattn_f16x3(const float* q, const float* k, const float* v, float* out, int nb, int nq,
                                      int nk, int heads, int dh, int ldq, int ldk, int ldv, int ldo, float scale,
                                      int32_t* status, cs_stream_t stream) {
  return attn16_dispatch<false>(q, k, v, out, nb, nq, nk, heads, dh, ldq, ldk, ldv, ldo, scale, status, stream);
}

// r5: the same kernel with the operands' power-of-two pre-scales given by the caller instead of the constant 16: a
// transformer block's q / k / v are Linear(LayerNorm(x)) -- bounded by the weights alone (|q_j| <= ||W_j||_2 (max|gamma| sqrt(C)
// + ||beta||_2)) -- so the host picks, once per checkpoint, the largest powers of two that keep q * scale * q_scale,
// k * k_scale and v * v_scale inside the fp16 range: no activation can raise CS_STATUS_F16X3_OVERFLOW here whatever the
// input, and operands keep the same relative precision at any weight scale (attention.py:179-218).
extern "C" int cs_attn_selfattn_f16x3_scaled(const float* q, const float* k, const float* v, float* out, int nb, int nq,
                                             int nk, int heads, int dh, int ldq, int ldk, int ldv, int ldo, float scale,
                                             float q_scale, float k_scale, float v_scale, int32_t* status,
                                             cs_stream_t stream) {
  if (!(q_scale > 0.f) || !(k_scale > 0.f) || !(v_scale > 0.f)) return CS_EINVAL;
  return attn16_dispatch<false>(q, k, v, out, nb, nq, nk, heads, dh, ldq, ldk, ldv, ldo, scale, status, stream, q_scale,
                                k_scale, v_scale);
}

extern "C" int cs_attn_selfattn_f16(const float* q, const float* k, const float* v, float* out, int nb, int nq,
                                    int nk, int heads, int dh, int ldq, int ldk, int ldv, int ldo, float scale,
                                    int32_t* status, cs_stream_t stream) {
  return attn16_dispatch<true>(q, k, v, out, nb, nq, nk, heads, dh, ldq, ldk, ldv, ldo, scale, status, stream);
}

// Workspace form of cs_attn_selfattn_f16x3: with `ws` (cs_attn_f16x3_ws_bytes(...) bytes, 16-byte aligned) K and V are
// split into their fp16 hi / lo tile images once per call and the kernel streams those (see attn_presplit_kernel);
// ws == NULL, or a shape the image path does not cover (ws_bytes == 0), runs the in-kernel split.  Same results, bit
// for bit.
extern "C" int64_t cs_attn_f16x3_ws_bytes(int nb, int nq, int nk, int heads, int dh) {
  if (nb <= 0 || nq <= 0 || nk <= 0 || heads <= 0 || dh <= 0) return 0;
  const int var = img_variant(nq, nk, dh);
  if (var == 8) return (int64_t)nb * heads * ((nk + 31) / 32) * AttnImg<8, 32>::TILE_BYTES;
  return 0;
}

static int attn_ws_impl(const float* q, const float* k, const float* v, float* out, int nb, int nq,
                       int nk, int heads, int dh, int ldq, int ldk, int ldv, int ldo, float scale,
                       int32_t* status, void* ws, cs_stream_t stream, float qs, float ks, float vs) {
  const int var = ws ? img_variant(nq, nk, dh) : 0;
  if (var == 0)
    return attn16_dispatch<false>(q, k, v, out, nb, nq, nk, heads, dh, ldq, ldk, ldv, ldo, scale, status, stream, qs, ks, vs);
  if (!q || !k || !v || !out || nb <= 0 || nq <= 0 || nk <= 0 || heads <= 0 || dh <= 0) return CS_EINVAL;
  if ((dh & 3) || (ldq & 3) || (ldk & 3) || (ldv & 3) || (ldo & 3)) return CS_EINVAL;
  if (ldq < heads * dh || ldk < heads * dh || ldv < heads * dh || ldo < heads * dh) return CS_EINVAL;
  if (((uintptr_t)q & 15) || ((uintptr_t)k & 15) || ((uintptr_t)v & 15) || ((uintptr_t)out & 15) || ((uintptr_t)ws & 15))
    return CS_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  return launch_attn16_img<8, 32, 4>(q, k, v, out, nb, nq, nk, heads, dh, ldq, ldk, ldv, ldo, scale, status, ws, s, qs, ks, vs);
}

extern "C" int cs_attn_selfattn_f16x3_ws(const float* q, const float* k, const float* v, float* out, int nb, int nq,
                                         int nk, int heads, int dh, int ldq, int ldk, int ldv, int ldo, float scale,
                                         int32_t* status, void* ws, cs_stream_t stream) {
  return attn_ws_impl(q, k, v, out, nb, nq, nk, heads, dh, ldq, ldk, ldv, ldo, scale, status, ws, stream, QK_SCALE, QK_SCALE,
                      QK_SCALE);
}

// r6 (ABI 18): the workspace form with the caller's operand pre-scales (see cs_attn_selfattn_f16x3_scaled) -- what an
// attention block fed by a GroupNorm takes (vqvae_modules.py:154-178 AttnBlock, openai_model_3d.py:360-366 AttentionBlock):
// its q / k / v are Conv1x1(GroupNorm(x)), bounded by the weights and the norm's affine parameters alone
// (cs_attnblock_static_scales), so no activation can leave the fp16 range there either.
extern "C" int cs_attn_selfattn_f16x3_ws_scaled(const float* q, const float* k, const float* v, float* out, int nb, int nq,
                                                int nk, int heads, int dh, int ldq, int ldk, int ldv, int ldo, float scale,
                                                float q_scale, float k_scale, float v_scale, int32_t* status, void* ws,
                                                cs_stream_t stream) {
  if (!(q_scale > 0.f) || !(k_scale > 0.f) || !(v_scale > 0.f)) return CS_EINVAL;
  return attn_ws_impl(q, k, v, out, nb, nq, nk, heads, dh, ldq, ldk, ldv, ldo, scale, status, ws, stream, q_scale, k_scale,
                      v_scale);
}
