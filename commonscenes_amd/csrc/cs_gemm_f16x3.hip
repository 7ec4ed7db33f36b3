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
#ifndef CS_ABLATE
#define CS_ABLATE 0   // debug: 1 = no global loads, 2 = no LDS stores, 4 = no LDS operand reads, 8 = no barrier
#endif

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

constexpr unsigned OOB = 0xFFF00000u;   // byte offset past every buffer (tensors are < 0xFFE00000 bytes): loads 0

// NB: take the builtin's result with `auto` and bit_cast the WHOLE vector -- element-wise extraction through
// an ext_vector_type copy makes hipcc (ROCm 7.2) narrow the load to one dword and replicate it.
__device__ __forceinline__ float4 ldg4(__amdgpu_buffer_rsrc_t rs, unsigned off) {
  const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
  const f32x4 f = __builtin_bit_cast(f32x4, v);
  return make_float4(f[0], f[1], f[2], f[3]);
}
__device__ __forceinline__ h8 ldh8(__amdgpu_buffer_rsrc_t rs, unsigned off) {
  const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
  return __builtin_bit_cast(h8, v);
}

// The K loop is ONE basic block: no data-dependent branches.  Out-of-image taps, rows past M, channels past
// cin and columns past cout are all expressed as an out-of-range buffer offset (the buffer unit returns 0),
// so the compiler is free to slot the next chunk's address math, fp32->fp16 hi/lo conversion and LDS stores
// into the issue gaps behind the current chunk's MFMAs (an in-order wave hides ~5 issues per 32-cycle MFMA).
template <int WMB, int WNB, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(256) void conv_gemm_f16x3_kernel(const CsConvGemm p, int M, int tiles_n,
                                                              int taps_hw, int kw_, int kg_per_tap,
                                                              unsigned x_bytes, unsigned w_bytes) {
  constexpr int BM = 32 * WMB * WAVES_M;
  constexpr int BN = 32 * WNB * WAVES_N;
  constexpr int A_SZ = BM * AROW;          // halves per A image
  constexpr int B_SZ = 2 * BN * 8;         // halves per B image: [2 k-groups][BN][8]
  constexpr int STAGE = 2 * A_SZ + 2 * B_SZ;
  constexpr int AUNITS = BM * 2;           // (row, 8-channel half-row) units
  constexpr int APT = (AUNITS + 255) / 256;
  constexpr int BUNITS = 2 * BN;           // 16-byte units per B image
  constexpr int BPT = (BUNITS + 255) / 256;
  constexpr int MAX_TAPS = 27;
  constexpr int DUMP = 2 * STAGE;          // 256 x 16 B scratch: where surplus loader lanes park their stores
  constexpr int ROWOFF = DUMP + 256 * 8;
  __shared__ __attribute__((aligned(16))) _Float16 smem[ROWOFF + 2 * BM * MAX_TAPS];
  unsigned* rowoff = reinterpret_cast<unsigned*>(smem + ROWOFF);   // [ntaps][BM] byte offsets into x, or OOB

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

  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t lrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w_lo, 0, w_bytes, 0x00020000);

  // ---- source-row table: rowoff[tap][row] = byte offset of the source row in x, or OOB when the tap falls
  // outside the (virtual, possibly upsampled) input or the row is past M.  Built once per workgroup.
  const int ntaps = p.kd * taps_hw;
  {
    const int vdin = p.din << p.ud, vhin = p.hin << p.uh, vwin = p.win << p.uw;
    for (int idx = tid; idx < BM * ntaps; idx += 256) {
      const int t = idx / BM;
      const int row = idx - t * BM;
      const int m = m0 + row;
      unsigned r = OOB;
      if (m < M) {
        int mm = m;
        const int ow = mm % p.wout;
        mm /= p.wout;
        const int oh = mm % p.hout;
        mm /= p.hout;
        const int od = mm % p.dout;
        const int n = mm / p.dout;
        const int kd_ = t / taps_hw;
        const int rem = t - kd_ * taps_hw;
        const int kh_ = rem / kw_;
        const int kwi = rem - kh_ * kw_;
        const int vd = od * p.sd - p.pd + kd_, vh = oh * p.sh - p.ph + kh_, vw = ow * p.sw - p.pw + kwi;
        if ((unsigned)vd < (unsigned)vdin && (unsigned)vh < (unsigned)vhin && (unsigned)vw < (unsigned)vwin)
          r = (unsigned)(((n * p.din + (vd >> p.ud)) * p.hin + (vh >> p.uh)) * p.win + (vw >> p.uw)) *
              (unsigned)(p.lda * 4);
      }
      rowoff[idx] = r;
    }
  }

  const int chunks_per_tap = kg_per_tap >> 1;   // cin16 / 16
  const int nk = ntaps * chunks_per_tap;

  // per-thread loader constants
  unsigned a_lds[APT], a_coff[APT], a_row[APT];
#pragma unroll
  for (int i = 0; i < APT; ++i) {
    const int u = tid + 256 * i;
    const bool ok = u < AUNITS;
    a_row[i] = ok ? (u >> 1) : 0;
    a_coff[i] = (u & 1) * 8;
    a_lds[i] = ok ? (unsigned)((u >> 1) * AROW + (u & 1) * 8) : (unsigned)(DUMP + tid * 8);
  }
  unsigned b_lds[BPT], b_goff[BPT];
#pragma unroll
  for (int i = 0; i < BPT; ++i) {
    const int u = tid + 256 * i;
    const int kg = u / BN;
    const int n = u - kg * BN;
    const bool ok = (u < BUNITS) && (n0 + n < p.cout);
    b_goff[i] = ok ? (unsigned)((kg * p.cout + n0 + n) * 16) : OOB;
    b_lds[i] = (u < BUNITS) ? (unsigned)(u * 8) : (unsigned)(DUMP + tid * 8);
  }

  float4 ra[APT][2];
  h8 rbh[BPT], rbl[BPT];

  auto load_chunk = [&](int tap, int cc) {   // cc = 16-channel chunk index; (tap, cc) past the end loads zeros
#pragma unroll
    for (int i = 0; i < APT; ++i) {
      const int c = cc * BKH + a_coff[i];
      const unsigned r = rowoff[tap * BM + a_row[i]];
      const unsigned o0 = (c < p.cin) ? r + (unsigned)c * 4u : OOB;
      const unsigned o1 = (c + 4 < p.cin) ? r + (unsigned)c * 4u + 16u : OOB;
      ra[i][0] = ldg4(xrs, o0);
      ra[i][1] = ldg4(xrs, o1);
    }
    const unsigned kbase = (unsigned)((tap * kg_per_tap + cc * 2) * p.cout) * 16u;
#pragma unroll
    for (int i = 0; i < BPT; ++i) {
      const unsigned o = (b_goff[i] == OOB) ? OOB : b_goff[i] + kbase;
      rbh[i] = ldh8(hrs, o);
      rbl[i] = ldh8(lrs, o);
    }
  };

  auto store_chunk = [&](int buf) {
    _Float16* s = smem + buf * STAGE;
#pragma unroll
    for (int i = 0; i < APT; ++i) {
      h8 hi, lo;
      split8(ra[i][0], ra[i][1], hi, lo);
      const bool ok = (tid + 256 * i) < AUNITS;
      _Float16* d = ok ? s : smem;      // surplus lanes write into the DUMP area (absolute offset)
      *reinterpret_cast<h8*>(d + a_lds[i]) = hi;
      *reinterpret_cast<h8*>(d + a_lds[i] + (ok ? A_SZ : 0)) = lo;
    }
#pragma unroll
    for (int i = 0; i < BPT; ++i) {
      const bool ok = (tid + 256 * i) < BUNITS;
      _Float16* d = ok ? s + 2 * A_SZ : smem;
      *reinterpret_cast<h8*>(d + b_lds[i]) = rbh[i];
      *reinterpret_cast<h8*>(d + b_lds[i] + (ok ? B_SZ : 0)) = rbl[i];
    }
  };

  f32x16 acc[WMB][WNB];
#pragma unroll
  for (int i = 0; i < WMB; ++i)
#pragma unroll
    for (int j = 0; j < WNB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  __syncthreads();   // rowoff table complete
  int tap = 0, cc = 0;
  load_chunk(tap, cc);
  store_chunk(0);
  __syncthreads();

  for (int kc = 0; kc < nk; ++kc) {
    const int buf = kc & 1;
    // K order: channel chunk OUTER, tap INNER (the 27 taps of one 16-channel chunk re-touch only this
    // tile's rows + halo, so they hit L1/L2).  The prefetch after the last chunk runs past cin16 and reads 0.
    if (++tap == ntaps) {
      tap = 0;
      ++cc;
    }
    if (!(CS_ABLATE & 1)) load_chunk(tap, cc);
    const _Float16* s = smem + ((CS_ABLATE & 4) ? 0 : buf * STAGE);
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
    if (!(CS_ABLATE & 2)) store_chunk(buf ^ 1);
    if (!(CS_ABLATE & 8)) __syncthreads();
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
  // buffer-descriptor extents: everything the loader may touch, and < 0xFFE00000 so OOB stays out of range
  const int64_t x_rows = (int64_t)p.nb * p.din * p.hin * p.win;
  const int64_t x_bytes = ((x_rows - 1) * p.lda + p.cin) * 4;
  const int64_t w_bytes = (int64_t)p.kd * p.kh * p.kw * kg_per_tap * p.cout * 16;
  if (x_bytes > 0xFFE00000LL || w_bytes > 0xFFE00000LL) return CS_EINVAL;
  CS_LAUNCH((conv_gemm_f16x3_kernel<WMB, WNB, WAVES_M, WAVES_N>), dim3((unsigned)nblk), dim3(256), 0,
            stream, p, M, tiles_n, p.kh * p.kw, p.kw, kg_per_tap, (unsigned)x_bytes, (unsigned)w_bytes);
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
  if (p.kd * p.kh * p.kw > 27) return CS_EINVAL;                                   // LDS row table extent
  if ((int64_t)p.nb * p.din * p.hin * p.win > 0x7fffffffLL) return CS_EINVAL;      // int32 row indices
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
