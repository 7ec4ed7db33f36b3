// Flash-style multi-head attention in fp32 on v_mfma_f32_32x32x2_f32 (gfx950).
//
// "Swapped" formulation so the softmax reduction axis is lane-local:
//   S^T[j][i] = sum_d K[j][d] Q[i][d]      (A = K tile from LDS, B = Q held in registers)
//   O^T[d][i] += sum_j V[j][d] P^T[j][i]    (A = V tile from LDS, B = P^T straight from the S^T
//                                            accumulator registers -- no cross-lane movement)
// In the 32x32 C/D layout lane (i = lane&31, half = lane>>5) holds S^T[j][i] for
// j = (r&3) + 8*(r>>2) + 4*half, r in [0,16).  MFMA k-step r of the PV product takes the key pair
// (j0(r), j0(r)+4): exactly what the two half-waves hold in register r.  Row max / row sum for
// query i need one shuffle with lane^32 and nothing else.
//
// Workgroup = 4 waves x 32 queries; K/V tiles of KT keys staged in LDS and shared by the 4 waves.
#include "cs_common.h"

namespace {

template <int DB, int KT>
__global__ __launch_bounds__(256) void attn_kernel(const float* __restrict__ q,
                                                   const float* __restrict__ k,
                                                   const float* __restrict__ v,
                                                   float* __restrict__ out, int nq, int nk,
                                                   int heads, int dh, int ldq, int ldk, int ldv,
                                                   int ldo, float scale, int qtiles) {
  constexpr int DP = 32 * DB;
  constexpr int LDKS = DP + 1;
  constexpr int LDVS = DP + 4;
  constexpr int JB = KT / 32;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ks = smem;                // [KT][LDKS]
  float* Vs = smem + KT * LDKS;    // [KT][LDVS]  (KT*LDKS*4 is a multiple of 16 for KT % 4 == 0)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int l31 = lane & 31;
  const int half = lane >> 5;

  int bid = blockIdx.x;
  const int qt = bid % qtiles;
  bid /= qtiles;
  const int h = bid % heads;
  const int b = bid / heads;

  const int q0 = qt * 128 + wave * 32;
  const int qi = min(q0 + l31, nq - 1);
  const float* qp = q + ((int64_t)b * nq + qi) * ldq + h * dh;
  const float* kb = k + (int64_t)b * nk * ldk + h * dh;
  const float* vb = v + (int64_t)b * nk * ldv + h * dh;

  // Q fragment: qreg[t] = Q[i][2t + half]
  float qreg[DP / 2];
#pragma unroll
  for (int t = 0; t < DP / 2; ++t) {
    const int d = 2 * t + half;
    qreg[t] = d < dh ? qp[d] * scale : 0.f;
  }

  // zero the padded LDS columns once
  for (int u = tid; u < KT * (DP - dh); u += 256) {
    const int j = u / (DP - dh);
    const int d = dh + (u - j * (DP - dh));
    Ks[j * LDKS + d] = 0.f;
    Vs[j * LDVS + d] = 0.f;
  }
  if (tid < KT * 4) Vs[(tid >> 2) * LDVS + DP + (tid & 3)] = 0.f;

  f32x16 oacc[DB];
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
  float mrun = -INFINITY;
  float lrun = 0.f;

  const int dh4 = dh >> 2;
  for (int kt0 = 0; kt0 < nk; kt0 += KT) {
    __syncthreads();  // previous tile fully consumed
    for (int u = tid; u < KT * dh4; u += 256) {
      const int j = u / dh4;
      const int c4 = u - j * dh4;
      float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
      if (kt0 + j < nk) {
        kv = *reinterpret_cast<const float4*>(kb + (int64_t)(kt0 + j) * ldk + c4 * 4);
        vv = *reinterpret_cast<const float4*>(vb + (int64_t)(kt0 + j) * ldv + c4 * 4);
      }
      float* kd = Ks + j * LDKS + c4 * 4;
      kd[0] = kv.x; kd[1] = kv.y; kd[2] = kv.z; kd[3] = kv.w;
      *reinterpret_cast<float4*>(Vs + j * LDVS + c4 * 4) = vv;
    }
    __syncthreads();

    // ---- S^T = K Q^T ----
    f32x16 sacc[JB];
#pragma unroll
    for (int jb = 0; jb < JB; ++jb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[jb][r] = 0.f;
      const float* kr = Ks + (jb * 32 + l31) * LDKS + half;
#pragma unroll
      for (int t = 0; t < DP / 2; ++t)
        sacc[jb] = __builtin_amdgcn_mfma_f32_32x32x2f32(kr[2 * t], qreg[t], sacc[jb], 0, 0, 0);
    }

    // ---- online softmax (per query i = lane&31) ----
    float mloc = -INFINITY;
#pragma unroll
    for (int jb = 0; jb < JB; ++jb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = kt0 + jb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (j >= nk) sacc[jb][r] = -INFINITY;
        mloc = fmaxf(mloc, sacc[jb][r]);
      }
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
    const float mnew = fmaxf(mrun, mloc);
    const float alpha = (mrun == -INFINITY) ? 0.f : expf(mrun - mnew);
    float psum = 0.f;
#pragma unroll
    for (int jb = 0; jb < JB; ++jb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = expf(sacc[jb][r] - mnew);
        sacc[jb][r] = pv;
        psum += pv;
      }
    lrun = lrun * alpha + psum;
    mrun = mnew;
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[d][r] *= alpha;

    // ---- O^T += V^T P^T ----
#pragma unroll
    for (int jb = 0; jb < JB; ++jb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = jb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const float* vr = Vs + j * LDVS + l31;
#pragma unroll
        for (int d = 0; d < DB; ++d)
          oacc[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(vr[32 * d], sacc[jb][r], oacc[d], 0, 0, 0);
      }
  }

  const float ltot = lrun + __shfl_xor(lrun, 32, 64);
  const float inv = 1.0f / ltot;
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

template <int DB, int KT>
int launch_attn(const float* q, const float* k, const float* v, float* out, int nb, int nq, int nk,
                int heads, int dh, int ldq, int ldk, int ldv, int ldo, float scale,
                hipStream_t s) {
  constexpr int DP = 32 * DB;
  const size_t smem = (size_t)KT * ((DP + 1) + (DP + 4)) * sizeof(float);
  const int qtiles = (nq + 127) / 128;
  const int64_t grid = (int64_t)qtiles * heads * nb;
  if (grid > 0x7fffffffLL) return CS_EINVAL;
  auto kern = attn_kernel<DB, KT>;
  if (smem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
  }
  CS_LAUNCH(kern, dim3((unsigned)grid), dim3(256), smem, s, q, k, v, out, nq, nk, heads,
                     dh, ldq, ldk, ldv, ldo, scale, qtiles);
  CS_CHECK_LAUNCH();
  return CS_OK;
}

}  // namespace

extern "C" int cs_attn_selfattn(const float* q, const float* k, const float* v, float* out, int nb,
                                int nq, int nk, int heads, int dh, int ldq, int ldk, int ldv,
                                int ldo, float scale, cs_stream_t stream) {
  if (!q || !k || !v || !out || nb <= 0 || nq <= 0 || nk <= 0 || heads <= 0 || dh <= 0)
    return CS_EINVAL;
  if ((dh & 3) || (ldq & 3) || (ldk & 3) || (ldv & 3) || (ldo & 3)) return CS_EINVAL;
  if (ldq < heads * dh || ldk < heads * dh || ldv < heads * dh || ldo < heads * dh)
    return CS_EINVAL;
  if (((uintptr_t)q & 15) || ((uintptr_t)k & 15) || ((uintptr_t)v & 15) || ((uintptr_t)out & 15))
    return CS_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (dh <= 32) return launch_attn<1, 64>(q, k, v, out, nb, nq, nk, heads, dh, ldq, ldk, ldv, ldo, scale, s);
  if (dh <= 64) return launch_attn<2, 64>(q, k, v, out, nb, nq, nk, heads, dh, ldq, ldk, ldv, ldo, scale, s);
  if (dh <= 96) return launch_attn<3, 64>(q, k, v, out, nb, nq, nk, heads, dh, ldq, ldk, ldv, ldo, scale, s);
  if (dh <= 128) return launch_attn<4, 64>(q, k, v, out, nb, nq, nk, heads, dh, ldq, ldk, ldv, ldo, scale, s);
  if (dh <= 256) return launch_attn<8, 32>(q, k, v, out, nb, nq, nk, heads, dh, ldq, ldk, ldv, ldo, scale, s);
  return CS_EINVAL;
}
