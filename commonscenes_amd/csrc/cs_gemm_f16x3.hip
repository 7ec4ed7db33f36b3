// Implicit-GEMM conv3d / linear with fp32 operands carried as fp16 hi/lo pairs on the fp16 MFMA
// (v_mfma_f32_32x32x16_f16) -- CS_MATH_F16X3.
//
// Why: gfx950's fp32-input MFMA runs at the vector rate (157 TF); its fp16 MFMA runs at 2.5 PF.  A fp32
// value v*2^s split as hi = fp16(v'), lo = fp16(v' - hi) keeps 22 mantissa bits, every fp16 x fp16 product is
// exact in the fp32 accumulator, and  a.w ~= a_hi.w_hi + a_hi.w_lo + a_lo.w_hi  drops only the 2^-22 term.
// Three fp16 MFMAs per K=16 step cost 96 SIMD cycles against 512 for the fp32-input MFMA: a 5.3x higher
// matrix-pipe ceiling at ~fp32 accuracy (measured error vs fp64 is reported by tests/test_f16x3_gpu.py).
//
// Weights are split offline (cs_pack_weight_f16x3), activations on the fly in the loader (scale 2^6).
// Tile BM x BN x 16, 4 waves; LDS holds [A_hi | A_lo | B_hi | B_lo] per stage, two stages (one barrier per
// K-chunk).  LDS images are MFMA-fragment shaped so every operand read is one conflict-free ds_read_b128:
//   A: [m][16 halves + 8 pad]  (48-byte rows: 16 consecutive rows hit 16 distinct 16-byte slots)
//   B: [k/8][n][8 halves]      (32 consecutive n = 512 contiguous bytes)
// fp16 32x32x16 operand map: lane l holds row/col l&31 and k = 8*(l>>5) .. 8*(l>>5)+7; C/D as for fp32.
#include "cs_common.h"

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

constexpr int BKH = 16;       // K elements per chunk
constexpr int AROW = 24;      // halves per LDS A row (16 data + 8 pad)
constexpr float A_SCALE = 64.0f;

__device__ __forceinline__ void split8(const float4& x, const float4& y, h8& hi, h8& lo) {
  const float v[8] = {x.x * A_SCALE, x.y * A_SCALE, x.z * A_SCALE, x.w * A_SCALE,
                      y.x * A_SCALE, y.y * A_SCALE, y.z * A_SCALE, y.w * A_SCALE};
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const _Float16 h = (_Float16)v[i];
    hi[i] = h;
    lo[i] = (_Float16)(v[i] - (float)h);
  }
}

template <int WMB, int WNB, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(256) void conv_gemm_f16x3_kernel(const CsConvGemm p, int M, int tiles_n,
                                                              int taps_hw, int kw_, int kg_per_tap) {
  constexpr int BM = 32 * WMB * WAVES_M;
  constexpr int BN = 32 * WNB * WAVES_N;
  constexpr int A_SZ = BM * AROW;          // halves per A image
  constexpr int B_SZ = 2 * BN * 8;         // halves per B image: [2 k-groups][BN][8]
  constexpr int STAGE = 2 * A_SZ + 2 * B_SZ;
  constexpr int AUNITS = BM * 2;           // (row, 8-channel half-row) units
  constexpr int APT = (AUNITS + 255) / 256;
  constexpr int BUNITS = 2 * BN;           // 16-byte units per B image
  constexpr int BPT = (BUNITS + 255) / 256;
  __shared__ __attribute__((aligned(16))) _Float16 smem[2 * STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int l31 = lane & 31;
  const int half = lane >> 5;
  const int wm0 = (wave / WAVES_N) * (32 * WMB);
  const int wn0 = (wave % WAVES_N) * (32 * WNB);

  int tile;
  {
    const int nblk = gridDim.x, b = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7, xcd = b & 7, within = b >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
  }
  const int tn = tile % tiles_n;
  const int tm = tile / tiles_n;
  const int m0 = tm * BM;
  const int n0 = tn * BN;

  // ---- per-thread A row bookkeeping: unit u = tid + 256*i -> row u>>1, channel octet u&1 ----
  int id0[APT], ih0[APT], iw0[APT];
  int64_t nbase[APT];
  bool rvalid[APT];
  const int vdin = p.din << p.ud, vhin = p.hin << p.uh, vwin = p.win << p.uw;
#pragma unroll
  for (int i = 0; i < APT; ++i) {
    const int u = tid + 256 * i;
    const int row = u >> 1;
    const int m = m0 + row;
    rvalid[i] = (u < AUNITS) && (m < M);
    int mm = rvalid[i] ? m : 0;
    const int ow = mm % p.wout;
    mm /= p.wout;
    const int oh = mm % p.hout;
    mm /= p.hout;
    const int od = mm % p.dout;
    const int n = mm / p.dout;
    id0[i] = od * p.sd - p.pd;
    ih0[i] = oh * p.sh - p.ph;
    iw0[i] = ow * p.sw - p.pw;
    nbase[i] = (int64_t)n * p.din * p.hin * p.win;
  }

  const int ntaps = p.kd * taps_hw;
  const int chunks_per_tap = kg_per_tap >> 1;   // cin16 / 16
  const int nk = ntaps * chunks_per_tap;
  const h8* wh = reinterpret_cast<const h8*>(p.w);
  const h8* wl = reinterpret_cast<const h8*>(p.w_lo);

  float4 ra[APT][2];
  h8 rbh[BPT], rbl[BPT];

  auto load_chunk = [&](int tap, int cc) {   // cc = chunk index within the tap (16 channels each)
    const int kd_ = tap / taps_hw;
    const int rem = tap - kd_ * taps_hw;
    const int kh_ = rem / kw_;
    const int kwi = rem - kh_ * kw_;
#pragma unroll
    for (int i = 0; i < APT; ++i) {
      const int u = tid + 256 * i;
      const int c = cc * BKH + (u & 1) * 8;
      const int vd = id0[i] + kd_, vh = ih0[i] + kh_, vw = iw0[i] + kwi;
      const bool ok = rvalid[i] && (unsigned)vd < (unsigned)vdin && (unsigned)vh < (unsigned)vhin &&
                      (unsigned)vw < (unsigned)vwin;
      float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
      if (ok) {
        const int64_t srow =
            nbase[i] + ((int64_t)(vd >> p.ud) * p.hin + (vh >> p.uh)) * p.win + (vw >> p.uw);
        const float* src = p.x + srow * p.lda + c;
        if (c < p.cin) v0 = *reinterpret_cast<const float4*>(src);
        if (c + 4 < p.cin) v1 = *reinterpret_cast<const float4*>(src + 4);
      }
      ra[i][0] = v0;
      ra[i][1] = v1;
    }
    const int64_t kg0 = (int64_t)tap * kg_per_tap + cc * 2;
#pragma unroll
    for (int i = 0; i < BPT; ++i) {
      const int u = tid + 256 * i;
      const int kg = u / BN;
      const int n = u - kg * BN;
      h8 vh_ = {0, 0, 0, 0, 0, 0, 0, 0}, vl_ = vh_;
      if (u < BUNITS && n0 + n < p.cout) {
        const int64_t off = (kg0 + kg) * p.cout + n0 + n;
        vh_ = wh[off];
        vl_ = wl[off];
      }
      rbh[i] = vh_;
      rbl[i] = vl_;
    }
  };

  auto store_chunk = [&](int buf) {
    _Float16* s = smem + buf * STAGE;
#pragma unroll
    for (int i = 0; i < APT; ++i) {
      const int u = tid + 256 * i;
      if (u < AUNITS) {
        h8 hi, lo;
        split8(ra[i][0], ra[i][1], hi, lo);
        const int off = (u >> 1) * AROW + (u & 1) * 8;
        *reinterpret_cast<h8*>(s + off) = hi;
        *reinterpret_cast<h8*>(s + A_SZ + off) = lo;
      }
    }
#pragma unroll
    for (int i = 0; i < BPT; ++i) {
      const int u = tid + 256 * i;
      if (u < BUNITS) {
        *reinterpret_cast<h8*>(s + 2 * A_SZ + u * 8) = rbh[i];
        *reinterpret_cast<h8*>(s + 2 * A_SZ + B_SZ + u * 8) = rbl[i];
      }
    }
  };

  f32x16 acc[WMB][WNB];
#pragma unroll
  for (int i = 0; i < WMB; ++i)
#pragma unroll
    for (int j = 0; j < WNB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  int tap = 0, cc = 0;
  load_chunk(tap, cc);
  store_chunk(0);
  __syncthreads();

  for (int kc = 0; kc < nk; ++kc) {
    const int buf = kc & 1;
    const bool more = (kc + 1) < nk;
    if (more) {
      if (++cc == chunks_per_tap) {
        cc = 0;
        ++tap;
      }
      load_chunk(tap, cc);
    }
    const _Float16* s = smem + buf * STAGE;
    h8 ah[WMB], al[WMB];
#pragma unroll
    for (int i = 0; i < WMB; ++i) {
      const int off = (wm0 + 32 * i + l31) * AROW + 8 * half;
      ah[i] = *reinterpret_cast<const h8*>(s + off);
      al[i] = *reinterpret_cast<const h8*>(s + A_SZ + off);
    }
#pragma unroll
    for (int j = 0; j < WNB; ++j) {
      const int off = (half * BN + wn0 + 32 * j + l31) * 8;
      const h8 bh = *reinterpret_cast<const h8*>(s + 2 * A_SZ + off);
      const h8 bl = *reinterpret_cast<const h8*>(s + 2 * A_SZ + B_SZ + off);
#pragma unroll
      for (int i = 0; i < WMB; ++i) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh, acc[i][j], 0, 0, 0);
      }
    }
    if (more) store_chunk(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue (identical contract to the fp32 kernel, after undoing the operand scales) ----
#pragma unroll
  for (int j = 0; j < WNB; ++j) {
    const int n = n0 + wn0 + 32 * j + l31;
    const bool nok = n < p.cout;
    const float bias = (nok && p.bias) ? p.bias[n] : 0.f;
    const float sc = (nok && p.scale) ? p.scale[n] : 1.f;
    const float sh = (nok && p.shift) ? p.shift[n] : 0.f;
#pragma unroll
    for (int i = 0; i < WMB; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * half;
        const int m = m0 + row;
        if (nok && m < M) {
          float v = acc[i][j][r] * p.acc_scale + bias;
          if (p.scale) v = v * sc + sh;
          if (p.rowvec) v += p.rowvec[(int64_t)(m / p.rv_rows) * p.ldrv + n];
          v = cs_act(v, p.act);
          if (p.res) v += p.res[(int64_t)m * p.ldr + n];
          p.out[(int64_t)m * p.ldo + n] = v;
        }
      }
    }
  }
}

template <int WMB, int WNB, int WAVES_M, int WAVES_N>
int launch16(const CsConvGemm& p, int M, hipStream_t stream) {
  constexpr int BM = 32 * WMB * WAVES_M;
  constexpr int BN = 32 * WNB * WAVES_N;
  const int tiles_m = (M + BM - 1) / BM;
  const int tiles_n = (p.cout + BN - 1) / BN;
  const int64_t nblk = (int64_t)tiles_m * tiles_n;
  if (nblk > 0x7fffffffLL) return CS_EINVAL;
  const int kg_per_tap = ((p.cin + 15) / 16) * 2;
  CS_LAUNCH((conv_gemm_f16x3_kernel<WMB, WNB, WAVES_M, WAVES_N>), dim3((unsigned)nblk), dim3(256), 0,
            stream, p, M, tiles_n, p.kh * p.kw, p.kw, kg_per_tap);
  CS_CHECK_LAUNCH();
  return CS_OK;
}

__global__ __launch_bounds__(256) void pack_f16x3_kernel(const float* __restrict__ w, _Float16* __restrict__ wh,
                                                         _Float16* __restrict__ wl, int cout, int cin, int taps,
                                                         int kg_per_tap, float scale) {
  const int64_t total = (int64_t)taps * kg_per_tap * cout * 8;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int j = (int)(i & 7);
    int64_t t = i >> 3;
    const int n = (int)(t % cout);
    t /= cout;
    const int kg = (int)(t % kg_per_tap);
    const int tap = (int)(t / kg_per_tap);
    const int c = kg * 8 + j;
    float v = 0.f;
    if (c < cin) v = w[((int64_t)n * cin + c) * taps + tap] * scale;
    const _Float16 h = (_Float16)v;
    wh[i] = h;
    wl[i] = (_Float16)(v - (float)h);
  }
}

}  // namespace

// called from cs_conv_gemm (cs_gemm.hip) when desc->math == CS_MATH_F16X3; arguments already validated
int cs_conv_gemm_f16x3_dispatch(const CsConvGemm& p, int M, int tile, hipStream_t s) {
  if (!p.w_lo || !(p.acc_scale > 0.f)) return CS_EINVAL;
  if (((uintptr_t)p.w & 15) || ((uintptr_t)p.w_lo & 15)) return CS_EINVAL;
  switch (tile) {
    case 1: return launch16<2, 2, 2, 2>(p, M, s);
    case 2: return launch16<1, 7, 4, 1>(p, M, s);
    case 3: return launch16<1, 1, 2, 2>(p, M, s);
    default: return CS_EINVAL;
  }
}

extern "C" int cs_pack_weight_f16x3(const float* w_torch, void* w_hi, void* w_lo, int cout, int cin, int taps,
                                    float scale, cs_stream_t stream) {
  if (!w_torch || !w_hi || !w_lo || cout <= 0 || cin <= 0 || taps <= 0 || !(scale > 0.f)) return CS_EINVAL;
  const int kg_per_tap = ((cin + 15) / 16) * 2;
  const int64_t total = (int64_t)taps * kg_per_tap * cout * 8;
  CS_LAUNCH(pack_f16x3_kernel, dim3(cs_grid_for(total, 256, 256 * 32)), dim3(256), 0, (hipStream_t)stream,
            w_torch, (_Float16*)w_hi, (_Float16*)w_lo, cout, cin, taps, kg_per_tap, scale);
  CS_CHECK_LAUNCH();
  return CS_OK;
}
