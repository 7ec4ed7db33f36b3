// Implicit-GEMM conv3d / linear on fp32-input MFMA (v_mfma_f32_32x32x2_f32) for gfx950.
//
// One kernel family covers every dense contraction on the CommonScenes shape path:
//   3x3x3 conv (stride 1 / (1,2,2), optional fused nearest upsample), 1x1x1 conv, nn.Linear.
// GEMM view: M = nb*dout*hout*wout output voxels (or tokens), N = cout, K = taps*cin.
// Activations are NDHWC so a K-chunk of one tap is a contiguous run of channels;
// weights are pre-laid-out [tap][cin][cout] so a B tile row is a contiguous run of cout.
//
// Workgroup = 256 threads = 4 waves, one per SIMD.  Tile BM x BN x 16, register-prefetched and
// double-buffered in LDS (one barrier per K-chunk).  A is stored transposed in LDS ([k][m]) so
// that both MFMA operand reads are 32 consecutive words (conflict free for ds_read_b32).
// MFMA operand maps (cdna_hip_programming.md section 3):
//   A[i = lane&31][k = lane>>5], B[k = lane>>5][j = lane&31],
//   D: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5), r in [0,16).
#include <cstdlib>
#include <atomic>
#include <cstring>
#include "cs_f16x3.h"

namespace {

constexpr int BK = 16;

template <int WMB, int WNB, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(256) void conv_gemm_f32_kernel(const CsConvGemm p, int M,
                                                            int tiles_n, int taps_hw, int kw_) {
  constexpr int BM = 32 * WMB * WAVES_M;
  constexpr int BN = 32 * WNB * WAVES_N;
  constexpr int LDAS = BM + 4;
  constexpr int LDBS = BN + 4;
  constexpr int RPT = (BM + 63) / 64;        // A rows per thread
  constexpr int BUNITS = BK * BN / 4;        // float4 units in a B tile
  constexpr int BPT = (BUNITS + 255) / 256;  // B float4 per thread
  constexpr int A_SZ = BK * LDAS;
  constexpr int B_SZ = BK * LDBS;
  __shared__ __attribute__((aligned(16))) float smem[2 * A_SZ + 2 * B_SZ];
  float* As = smem;
  float* Bs = smem + 2 * A_SZ;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int l31 = lane & 31;
  const int half = lane >> 5;
  const int wm0 = (wave / WAVES_N) * (32 * WMB);
  const int wn0 = (wave % WAVES_N) * (32 * WNB);

  // XCD-aware tile mapping: block b runs on XCD b%8; give each XCD a contiguous run of tiles so
  // neighbouring tiles (same A rows / same weight panel) share that XCD's L2.
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

  // ---- per-thread A row bookkeeping ----
  const int kq = tid & 3;
  int id0[RPT], ih0[RPT], iw0[RPT];
  int64_t nbase[RPT];
  bool rvalid[RPT];
  const int vdin = p.din << p.ud, vhin = p.hin << p.uh, vwin = p.win << p.uw;
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const int row = (tid >> 2) + 64 * i;
    const int m = m0 + row;
    rvalid[i] = (row < BM) && (m < M);
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

  const int kchunks_per_tap = (p.cin + BK - 1) / BK;
  const int ntaps = p.kd * taps_hw;
  const int nk = ntaps * kchunks_per_tap;

  float4 ra[RPT];
  float4 rb[BPT];

  auto load_chunk = [&](int tap, int c0) {
    const int kd_ = tap / taps_hw;
    const int rem = tap - kd_ * taps_hw;
    const int kh_ = rem / kw_;
    const int kwi = rem - kh_ * kw_;
    const int c = c0 + kq * 4;
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const int vd = id0[i] + kd_, vh = ih0[i] + kh_, vw = iw0[i] + kwi;
      const bool ok = rvalid[i] && (unsigned)vd < (unsigned)vdin && (unsigned)vh < (unsigned)vhin &&
                      (unsigned)vw < (unsigned)vwin && c < p.cin;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ok) {
        const int64_t srow =
            nbase[i] + ((int64_t)(vd >> p.ud) * p.hin + (vh >> p.uh)) * p.win + (vw >> p.uw);
        v = *reinterpret_cast<const float4*>(p.x + srow * p.lda + c);
      }
      ra[i] = v;
    }
    const float* wt = p.w + ((int64_t)tap * p.cin) * p.ldw;
#pragma unroll
    for (int i = 0; i < BPT; ++i) {
      const int u = tid + 256 * i;
      const int krow = u / (BN / 4);
      const int c4 = u - krow * (BN / 4);
      const int kk = c0 + krow;
      const int n = n0 + c4 * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (u < BUNITS && kk < p.cin && n < p.ldw)
        v = *reinterpret_cast<const float4*>(wt + (int64_t)kk * p.ldw + n);
      rb[i] = v;
    }
  };

  auto store_chunk = [&](int buf) {
    float* a = As + buf * A_SZ;
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const int row = (tid >> 2) + 64 * i;
      if (row < BM) {
        a[(kq * 4 + 0) * LDAS + row] = ra[i].x;
        a[(kq * 4 + 1) * LDAS + row] = ra[i].y;
        a[(kq * 4 + 2) * LDAS + row] = ra[i].z;
        a[(kq * 4 + 3) * LDAS + row] = ra[i].w;
      }
    }
    float* b = Bs + buf * B_SZ;
#pragma unroll
    for (int i = 0; i < BPT; ++i) {
      const int u = tid + 256 * i;
      if (u < BUNITS) {
        const int krow = u / (BN / 4);
        const int c4 = u - krow * (BN / 4);
        *reinterpret_cast<float4*>(b + krow * LDBS + c4 * 4) = rb[i];
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

  int tap = 0, c0 = 0;
  load_chunk(tap, c0);
  store_chunk(0);
  __syncthreads();

  for (int kc = 0; kc < nk; ++kc) {
    const int buf = kc & 1;
    const bool more = (kc + 1) < nk;
    if (more) {
      // K order: channel chunk OUTER, tap INNER -- the 27 taps of one 16-channel chunk re-touch only this
      // tile's rows + halo, so they hit L1/L2 instead of re-streaming the tensor from MALL/HBM per tap.
      if (++tap == ntaps) {
        tap = 0;
        c0 += BK;
      }
      load_chunk(tap, c0);
    }
    const float* a = As + buf * A_SZ + wm0 + l31;
    const float* b = Bs + buf * B_SZ + wn0 + l31;
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      float av[WMB], bv[WNB];
#pragma unroll
      for (int i = 0; i < WMB; ++i) av[i] = a[(2 * kk + half) * LDAS + 32 * i];
#pragma unroll
      for (int j = 0; j < WNB; ++j) bv[j] = b[(2 * kk + half) * LDBS + 32 * j];
#pragma unroll
      for (int i = 0; i < WMB; ++i)
#pragma unroll
        for (int j = 0; j < WNB; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
    if (more) store_chunk(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue ----
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
          float v = acc[i][j][r] + bias;
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
int launch(const CsConvGemm& p, int M, hipStream_t stream) {
  constexpr int BM = 32 * WMB * WAVES_M;
  constexpr int BN = 32 * WNB * WAVES_N;
  const int tiles_m = (M + BM - 1) / BM;
  const int tiles_n = (p.cout + BN - 1) / BN;
  const int64_t nblk = (int64_t)tiles_m * tiles_n;
  if (nblk > 0x7fffffffLL) return CS_EINVAL;
  CS_LAUNCH((conv_gemm_f32_kernel<WMB, WNB, WAVES_M, WAVES_N>), dim3((unsigned)nblk),
                     dim3(256), 0, stream, p, M, tiles_n, p.kh * p.kw, p.kw);
  CS_CHECK_LAUNCH();
  return CS_OK;
}

__global__ void relayout_weight_kernel(const float* __restrict__ w, float* __restrict__ o, int cout,
                                       int cin, int taps, int cin_pad, int ldw) {
  // w: (cout, cin, taps) torch order; o: [tap][cin_pad][ldw]
  const int64_t total = (int64_t)taps * cin_pad * ldw;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int n = (int)(i % ldw);
    const int64_t t = i / ldw;
    const int c = (int)(t % cin_pad);
    const int tap = (int)(t / cin_pad);
    float v = 0.f;
    if (n < cout && c < cin) v = w[((int64_t)n * cin + c) * taps + tap];
    o[i] = v;
  }
}

}  // namespace

int cs_conv_gemm_f16x3_dispatch(const CsConvGemm& p, int M, int tile, int splits, hipStream_t s, int omap_f = 0,
                                int omap_p = 0, const void* const* cls_w = nullptr, const void* const* cls_w_lo = nullptr,
                                const float* cls_acc = nullptr, int ncls = 0,
                                const CsFuseK* fuse = nullptr, int u_base = 0, int u_count = 0);   // cs_gemm_f16x3.hip
bool cs_f16x3_slab4_ok(const CsConvGemm& p, int tile, int splits);                      // cs_gemm_f16x3.hip
// (tile code 5 -- r2's persistent ping-pong pointwise GEMM, cs_gemm_pw.hip -- was removed in r6: measured a loser in r2 at
// every shape (NOTES section 4.3: 128-row tiles double the weight DMA per flop, one wave per SIMD keeps the matrix pipe ~55 %
// busy), never selected since, and VERDICT r5 asked for it to earn its place or go; the code refuses the tile code)
bool cs_kw_gemm_applicable(const CsConvGemm& p, int64_t M);                                          // cs_gemm_kw.hip
int cs_kw_gemm_f16x3_launch(const CsConvGemm& p, int M, hipStream_t s);
static int device_cus();

namespace {

// Split-K heuristic: F16X3 GEMMs on the 224-column tile whose output tiles would leave CUs idle (small and medium
// batches) and whose K loop is long enough to share.  Tuned on the 1- / 4- / 7-object shapes (tools/gemm_smallm.py,
// tools/plan_ab.sh, profiles/r02_smallm_sweep.txt): the 128x224 tile runs two workgroups per CU, so the slice count is
// the largest POWER OF TWO that keeps tiles x slices <= 512 (odd counts fight the XCD-aware tile map: 11 slices of a
// 48-tile GEMM ran 247 us, 8 slices 178), at least 8 K chunks per slice, at most 32 slices; K loops under 64 chunks
// (the C x C token GEMMs) are not split -- the 64x64 tile fills the chip better there.
// Large batches: a 3x3x3 conv whose 256x224 tiles fill the chip's 256 CUs unevenly (the 4^3 level has three column
// tiles: 64 x 3 = 192 workgroups at 32 objects, a quarter of the CUs idle) is cut into FOUR K slices -- 768 workgroups,
// three even rounds -- whenever it has an odd number of column tiles and a long K loop.  The rule looks at K and N only
// (not at the batch beyond "large"), so a shard of a batch and the whole batch add their products in the same order.
// (r3: also with pre-split activations, a_format = 1 -- the GroupNorm producers feed exactly these convs in that form,
// so the r2 rule "a_format == 0" had switched the cut off on the model path; -DCS_SPLIT4_FP32_ONLY restores it for A/B.)
bool split4_large(const CsConvGemm& p, int64_t M) {
#ifdef CS_SPLIT4_FP32_ONLY
  if (p.a_format != 0) return false;
#endif
  return p.kd == 3 && p.kh == 3 && p.kw == 3 && p.sd == 1 && p.sh == 1 && p.sw == 1 && !(p.ud | p.uh | p.uw) &&
         ((p.cout / 224) & 1) && p.cout / 224 >= 3 && (p.cin + 15) / 16 >= 16 &&
         ((M + 255) / 256) * (int64_t)(p.cout / 224) >= 192;
}

int plan_splitk(const CsConvGemm& p, int64_t M) {
  if (p.math != CS_MATH_F16X3 || p.act == CS_ACT_GEGLU || p.tile != 0 || p.cout % 224) return 1;
  const int64_t wgs = ((M + 127) / 128) * (p.cout / 224);
  const int64_t nk = (int64_t)p.kd * p.kh * p.kw * ((p.cin + 15) / 16);
#ifndef CS_NO_SPLIT4
  if (split4_large(p, M)) return 4;
#endif
#ifndef CS_PLAN_R1
  if (wgs >= 384) return 1;
#else
  if (wgs >= 160) return 1;
#endif
#ifdef CS_PLAN_R1      // round-1 rule, for A/B timing builds (CS_EXTRA_HIPCC_FLAGS=-DCS_PLAN_R1)
  int64_t s1 = (256 + wgs - 1) / wgs;
  if (s1 > nk / 16) s1 = nk / 16;
  if (s1 > 32) s1 = 32;
  return s1 < 2 ? 1 : (int)s1;
#endif
  if (nk < 64) return 1;
  // r3: ANY slice count -- the largest s with tiles x s <= 512, i.e. exactly one round of the chip's 2-per-CU workgroup
  // slots (at most 32 slices, at least 8 K chunks each).  The power-of-two rule left slots empty (48 tiles x 8 = 384) or
  // spilled into a second, mostly empty round (84 tiles x 8 = 672 under the long-K exception): at 7 objects the 4^3-level
  // convs take 6 slices (244 / 452 us against 274 / 500 for 8), at 4 objects 10 (155 / 277 against 163 / 299), at 14
  // objects 3 (455 / 867 against 499 / 936 for 4) -- tools/gemm_smallm.py with SM_SPLITS, profiles/r03_ai_splitk_any.txt.
  // (r2's "odd counts fight the tile map" was 48 x 11 = 528 > 512: a second round, not the oddness.)
  // CS_PLAN_POW2=1: the previous rule (largest power of two, up to 704 workgroups for K loops of >= 1024 chunks), A/B runs.
  // Short K loops (the 1x1x1 / token GEMMs: <= 168 chunks) keep the power-of-two rule: more, shorter slices only add
  // partial-tile traffic there (1792 -> 448 at 2048 rows: 8 slices 40.6 us, 16 slices 43.1).
  // r4 (tools/gemm_tok_smallm.py, profiles/r04_tok_smallm_b{2,14,64}.txt): 1-tap GEMMs with K < 2048 are better off
  // UNSPLIT on the 128x128 / 64x64 tiles at every batch measured -- 1792 -> 448 at 14336 rows: 82.0 us on 128x128 against
  // 95.6 for two slices of 128x224, at 2048 rows 25.2 (64x64) against 35.6 (eight slices); 1120 -> 448: 65.3 vs 70.8 and
  // 25.2 vs 37.5 -- their K loop (<= 112 chunks) is too short to pay for partial tiles + the reduce launch.  Only the
  // 2688 -> 672 ff.net.2 (168 chunks) keeps its slices (66.2 vs 76.9 at 3584 rows, 28.8 vs 33.3 at 512).
  if (p.kd * p.kh * p.kw == 1 && nk < 128 && !cs_debug()->plan_pow2 && !cs_debug()->no_tok_rules) return 1;
  if (cs_debug()->plan_pow2 || !(p.kd == 3 && p.kh == 3 && p.kw == 3)) {
    const int64_t limit = nk >= 1024 ? 704 : 512;
    int64_t s = 1;
    while (2 * s * wgs <= limit && 2 * s <= 32 && 2 * s <= nk / 8) s *= 2;
    return (int)s;
  }
  int64_t s = 512 / wgs;
  // (r5, measured and not taken: a cap of 48 -- one object's 4^3 level is 12 output tiles: 32 slices = 192 workgroups of the
  // 256-row tile on 256 CUs, 42 slices = 252 -- ran 6.54 / 6.53 / 6.54 ms per one-object step against 6.47 / 6.47 / 6.45 at
  // 32: a third more partial-tile traffic for a quarter shorter K loops, profiles/r05_h_splitk_cap_ab.txt; CS_SPLITK_CAP
  // overrides for A/B runs)
  static const int64_t cap = [] {
    const char* e = getenv("CS_SPLITK_CAP");
    return (e && *e) ? atoll(e) : 32LL;
  }();
  if (s > cap) s = cap;
  if (s > nk / 8) s = nk / 8;
  if (s < 1) s = 1;
  {
    // the slab kernel cuts K in whole super-chunks (nine taps) and is taken only where that pads the slices by at most a
    // tenth (cs_conv_gemm_f16x3_dispatch: slab_slices_ok): the fewest slices of the same length (no empty last slice:
    // 84 super-chunks over 15 slices of 6 is 14 slices), stepping down to the next count that qualifies
    const int64_t nsc = 3LL * ((p.cin + 15) / 16);
    s = (nsc + (nsc + s - 1) / s - 1) / ((nsc + s - 1) / s);
    while (s > 1 && ((nsc + s - 1) / s) * s * 10 > nsc * 11) {
      --s;
      s = (nsc + (nsc + s - 1) / s - 1) / ((nsc + s - 1) / s);
    }
  }
  return s < 1 ? 1 : (int)s;
}

// sums the split-K partial tiles in slice order and applies the epilogue of conv_gemm_* (bias, BN scale/shift,
// row vector, activation, residual); one float4 of one output row per thread
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const CsConvGemm p, const float* __restrict__ ws, int M,
                                                            int splits) {
  const int n4 = p.cout >> 2;
  const int64_t total = (int64_t)M * n4;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int m = (int)(i / n4);
    const int n = (int)(i - (int64_t)m * n4) * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(ws + (int64_t)m * p.cout + n);
    for (int s = 1; s < splits; ++s) v += *reinterpret_cast<const f32x4*>(ws + ((int64_t)s * M + m) * p.cout + n);
    if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
    if (p.scale) v = v * *reinterpret_cast<const f32x4*>(p.scale + n) + *reinterpret_cast<const f32x4*>(p.shift + n);
    if (p.rowvec) v += *reinterpret_cast<const f32x4*>(p.rowvec + (int64_t)(m / p.rv_rows) * p.ldrv + n);
    if (p.act != CS_ACT_NONE) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = cs_act(v[e], p.act);
    }
    if (p.res) v += *reinterpret_cast<const f32x4*>(p.res + (int64_t)m * p.ldr + n);
    *reinterpret_cast<f32x4*>(p.out + (int64_t)m * p.ldo + n) = v;
  }
}

// splitk_reduce_kernel with the r4 epilogue outputs (CsConvGemm.gn_part / out_format): a workgroup owns SKR rows x 64
// columns -- 16 float4 column lanes x 16 row lanes, one row per thread (a first version gave a thread two rows: half the
// workgroups, 16 instead of 9.5 us per call at one object) -- so that it can (a) leave the fp64 (sum, sum of
// squares) of its rows per column for the GroupNorm that follows (statistics tile = SKR rows; at small batches nearly
// every GroupNorm input comes out of this kernel) and (b) write the interleaved operand pair, lanes 2t / 2t + 1 swapping a
// half as in the GEMM's own epilogue.  Same sums, same order per element as splitk_reduce_kernel.
constexpr int SKR = 16;
__global__ __launch_bounds__(256) void splitk_reduce_epi_kernel(const CsConvGemm p, const float* __restrict__ ws, int M,
                                                                int splits) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));
  __shared__ double sc[16][64];
  const int cbn = (p.cout + 63) / 64;
  const int rb = blockIdx.x / cbn, cbi = blockIdx.x - rb * cbn;
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int n = cbi * 64 + cl * 4;
  const bool nok = n < p.cout;
  const bool gstat = p.gn_part != nullptr, opair = p.out_format == 2;
  double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
  float oamax = 0.f;
#pragma unroll
  for (int rr = 0; rr < SKR / 16; ++rr) {
    const int m = rb * SKR + rl + 16 * rr;
    const bool ok = nok && m < M;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (ok) {
      v = *reinterpret_cast<const f32x4*>(ws + (int64_t)m * p.cout + n);
      for (int sl = 1; sl < splits; ++sl) v += *reinterpret_cast<const f32x4*>(ws + ((int64_t)sl * M + m) * p.cout + n);
      if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
      if (p.scale) v = v * *reinterpret_cast<const f32x4*>(p.scale + n) + *reinterpret_cast<const f32x4*>(p.shift + n);
      if (p.rowvec) v += *reinterpret_cast<const f32x4*>(p.rowvec + (int64_t)(m / p.rv_rows) * p.ldrv + n);
      if (p.act != CS_ACT_NONE) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = cs_act(v[e], p.act);
      }
      if (p.res) v += *reinterpret_cast<const f32x4*>(p.res + (int64_t)m * p.ldr + n);
      if (gstat) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const double d = (double)v[e];
          s[e] += d;
          q[e] += d * d;
        }
      }
    }
    if (opair) {      // uniform branch: every lane takes part in the half swap
      h4 hi, lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float o = v[e] * p.out_scale;
        oamax = fmaxf(oamax, fabsf(o));
        hi[e] = (_Float16)o;
        lo[e] = (_Float16)(o - (float)hi[e]);
      }
      const u32x2 H = __builtin_bit_cast(u32x2, hi), L = __builtin_bit_cast(u32x2, lo);
      const bool odd = cl & 1;
      const u32x2 send = odd ? H : L;
      u32x2 recv;
      recv[0] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)send[0], 0xB1, 0xF, 0xF, true);
      recv[1] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)send[1], 0xB1, 0xF, 0xF, true);
      u32x4 r;
      r[0] = odd ? recv[0] : H[0];
      r[1] = odd ? recv[1] : H[1];
      r[2] = odd ? L[0] : recv[0];
      r[3] = odd ? L[1] : recv[1];
      if (ok) {
        char* row = reinterpret_cast<char*>(p.out + (int64_t)m * p.ldo);
        *reinterpret_cast<u32x4*>(row + (n >> 4) * 64 + ((n & 8) ? 32 : 0) + ((n & 4) ? 16 : 0)) = r;
      }
    } else if (ok) {
      *reinterpret_cast<f32x4*>(p.out + (int64_t)m * p.ldo + n) = v;
    }
  }
  if (opair && p.status && oamax >= 65504.f) atomicOr(p.status, CS_STATUS_F16X3_OVERFLOW);
  if (gstat) {
    double tot[2][4];
#pragma unroll
    for (int which = 0; which < 2; ++which) {
      __syncthreads();
#pragma unroll
      for (int e = 0; e < 4; ++e) sc[rl][cl * 4 + e] = which ? q[e] : s[e];
      __syncthreads();
      if (rl == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          double t = 0.0;
          for (int r2 = 0; r2 < 16; ++r2) t += sc[r2][cl * 4 + e];
          tot[which][e] = t;
        }
      }
    }
    if (rl == 0 && nok) {
      double* o = p.gn_part + ((int64_t)rb * p.gn_ld + n) * 2;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o[2 * e] = tot[0][e];
        o[2 * e + 1] = tot[1][e];
      }
    }
  }
}

// r5: output transform + epilogue of the Winograd-W route, one TILE of R outputs per thread (R = 2: F(2,3), 4: F(4,3)):
// the thread reads its R + 2 position results once (each summed over its K slices in slice order), forms the R outputs
// (slice-order sums, then A^T m in a fixed order) and applies the epilogue row by row.  A workgroup
// = 16 tile lanes x 16 float4 column lanes = 16 R rows x 64 columns, which is also its GroupNorm statistics tile (fp64
// sums: a thread over its R rows in row order, then the 16 tile lanes in order).  The first version gave every output ROW a
// thread: each position result was fetched by up to four rows' threads and the kernel ran at ~3 TB/s of useful traffic.
template <int R>
__global__ __launch_bounds__(256) void wino_out_kernel(const CsConvGemm p, const float* __restrict__ ws, int M, int splits,
                                                       int tail_unit0, int tail_splits, int unit_cols) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));
  constexpr int P = R + 2;
  __shared__ double sc[16][64];
  const int cbn = (p.cout + 63) / 64;
  const int rb = blockIdx.x / cbn, cbi = blockIdx.x - rb * cbn;
  const int cl = threadIdx.x & 15, tl = threadIdx.x >> 4;
  const int n = cbi * 64 + cl * 4;
  const bool nok = n < p.cout;
  const int64_t Mt = M / R;                                    // tiles = rows of a position
  const int64_t t = (int64_t)rb * 16 + tl;
  const bool tok = nok && t < Mt;
  const bool gstat = p.gn_part != nullptr, opair = p.out_format == 2;
  f32x4 mq[P];
#pragma unroll
  for (int q = 0; q < P; ++q) mq[q] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (tok) {
    const int64_t pstride = Mt * p.cout, sstride = P * pstride;
    const float* b = ws + t * p.cout + n;
#pragma unroll
    for (int q = 0; q < P; ++q) {
      f32x4 a = *reinterpret_cast<const f32x4*>(b + q * pstride);
      // (r6: the (position, column tile) units from `tail_unit0` on came from the K-sliced tail launch: unit = q * tiles_n +
      // n / unit_cols -- a function of position and column only, so a sample's sums never depend on its place in the batch)
      const int nsl = (q * ((p.cout + unit_cols - 1) / unit_cols) + n / unit_cols >= tail_unit0) ? tail_splits : splits;
      for (int sl = 1; sl < nsl; ++sl) a += *reinterpret_cast<const f32x4*>(b + q * pstride + sl * sstride);
      mq[q] = a;
    }
  }
  f32x4 y[R];
  if constexpr (R == 4) {
    y[0] = mq[0];
    y[0] += mq[1];
    y[0] += mq[2];
    y[0] += mq[3];
    y[0] += mq[4];
    y[1] = mq[1] - mq[2];
    y[1] += 2.f * (mq[3] - mq[4]);
    y[2] = mq[1] + mq[2];
    y[2] += 4.f * (mq[3] + mq[4]);
    y[3] = mq[1] - mq[2];
    y[3] += 8.f * (mq[3] - mq[4]);
    y[3] += mq[5];
  } else {
    y[0] = mq[0];
    y[0] += mq[1];
    y[0] += mq[2];
    y[1] = mq[1];
    y[1] -= mq[2];
    y[1] -= mq[3];
  }
  double s[4] = {0, 0, 0, 0}, sq[4] = {0, 0, 0, 0};
  float oamax = 0.f;
#pragma unroll
  for (int e = 0; e < R; ++e) {
    const int64_t m = t * R + e;
    f32x4 v = y[e];
    if (tok) {
      if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
      if (p.scale) v = v * *reinterpret_cast<const f32x4*>(p.scale + n) + *reinterpret_cast<const f32x4*>(p.shift + n);
      if (p.rowvec) v += *reinterpret_cast<const f32x4*>(p.rowvec + (m / p.rv_rows) * p.ldrv + n);
      if (p.act != CS_ACT_NONE) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = cs_act(v[k], p.act);
      }
      if (p.res) v += *reinterpret_cast<const f32x4*>(p.res + m * p.ldr + n);
      if (gstat) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const double d = (double)v[k];
          s[k] += d;
          sq[k] += d * d;
        }
      }
    } else {
      v = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (opair) {      // uniform branch: every lane takes part in the half swap
      h4 hi, lo;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float o = v[k] * p.out_scale;
        oamax = fmaxf(oamax, fabsf(o));
        hi[k] = (_Float16)o;
        lo[k] = (_Float16)(o - (float)hi[k]);
      }
      const u32x2 H = __builtin_bit_cast(u32x2, hi), Lw = __builtin_bit_cast(u32x2, lo);
      const bool odd = cl & 1;
      const u32x2 send = odd ? H : Lw;
      u32x2 recv;
      recv[0] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)send[0], 0xB1, 0xF, 0xF, true);
      recv[1] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)send[1], 0xB1, 0xF, 0xF, true);
      u32x4 r;
      r[0] = odd ? recv[0] : H[0];
      r[1] = odd ? recv[1] : H[1];
      r[2] = odd ? Lw[0] : recv[0];
      r[3] = odd ? Lw[1] : recv[1];
      if (tok) {
        char* row = reinterpret_cast<char*>(p.out + m * p.ldo);
        *reinterpret_cast<u32x4*>(row + (n >> 4) * 64 + ((n & 8) ? 32 : 0) + ((n & 4) ? 16 : 0)) = r;
      }
    } else if (tok) {
      *reinterpret_cast<f32x4*>(p.out + m * p.ldo + n) = v;
    }
  }
  if (opair && p.status && oamax >= 65504.f) atomicOr(p.status, CS_STATUS_F16X3_OVERFLOW);
  if (gstat) {
    double tot[2][4];
#pragma unroll
    for (int which = 0; which < 2; ++which) {
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 4; ++k) sc[tl][cl * 4 + k] = which ? sq[k] : s[k];
      __syncthreads();
      if (tl == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          double a = 0.0;
          for (int r2 = 0; r2 < 16; ++r2) a += sc[r2][cl * 4 + k];
          tot[which][k] = a;
        }
      }
    }
    if (tl == 0 && nok) {
      double* o = p.gn_part + ((int64_t)rb * p.gn_ld + n) * 2;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        o[2 * k] = tot[0][k];
        o[2 * k + 1] = tot[1][k];
      }
    }
  }
}

// rows x columns of a tile code's output tile (0: no fixed tile -- the 512-row codes that may fall back)
void tile_dims(int tile, int& bm, int& bn) {
  switch (tile) {
    case 1: bm = 128; bn = 128; break;
    case 2: bm = 128; bn = 224; break;
    case 3: bm = 64; bn = 64; break;
    case 4: bm = 256; bn = 224; break;
    case 6: bm = 256; bn = 128; break;
    case 7: bm = 256; bn = 64; break;
    case 10: bm = 64; bn = 64; break;      // r5: the K-wave kernel (cs_gemm_kw.hip): its statistics tiles are its 64 rows
    default: bm = 0; bn = 0; break;
  }
}

}  // namespace

static int auto_tile(const CsConvGemm& p, int M, bool f16x3);
static bool up2_direct_batched(const CsConvGemm& d, int64_t m1, int ncls);
static int up2_batched_tile(const CsConvGemm& d, int64_t m1, int ncls);

// ONE rule for what a launch's epilogue can emit (CsConvGemm.gn_part / out_format): only the pipelined epilogue of the
// F16X3 tile kernels and the split-K reduce can, so every condition under which the kernel takes another path is
// checked here, on the host, for ALL tiles of the launch -- the hosts ask before they allocate, cs_conv_gemm re-checks.
extern "C" int cs_conv_gemm_epilogue_caps(const CsConvGemm* d, int32_t* gn_rows, int32_t* pair_ok) {
  if (!d) return CS_EINVAL;
  if (gn_rows) *gn_rows = 0;
  if (pair_ok) *pair_ok = 0;
  const CsConvGemm& p = *d;
  if (p.math != CS_MATH_F16X3 || p.nb <= 0 || p.cout <= 0 || p.dout <= 0 || p.hout <= 0 || p.wout <= 0) return CS_OK;
  const int64_t M64 = (int64_t)p.nb * p.dout * p.hout * p.wout;
  if (M64 > 0x7fffffffLL) return CS_OK;
  const int M = (int)M64;
  auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
  const int ocols = p.act == CS_ACT_GEGLU ? p.cout / 2 : p.cout;
  const bool vec = (p.cout % 4 == 0) && (p.ldo % 4 == 0) && al16(p.out) && (!p.bias || al16(p.bias)) &&
                   (!p.scale || (al16(p.scale) && al16(p.shift))) && (!p.rowvec || (p.ldrv % 4 == 0 && al16(p.rowvec))) &&
                   (!p.res || (p.ldr % 4 == 0 && al16(p.res)));
  if (!vec) return CS_OK;
  const bool pair_geom = (ocols % 8 == 0) && (p.ldo % 16 == 0) && (((uintptr_t)p.out & 63) == 0);
  int64_t rps = (int64_t)p.dout * p.hout * p.wout;           // output rows per sample
  int bm = 0, bn = 0;
  if (p.ud | p.uh | p.uw) {
    // folded Upsample conv (cs_conv_gemm_up2): only the one-launch direct-store route, tiles over the SOURCE rows
    const int64_t m1 = (int64_t)p.nb * p.din * p.hin * p.win;
    const int ncls = (p.ud ? 2 : 1) * (p.uh ? 2 : 1) * (p.uw ? 2 : 1);
    if (p.splitk > 1 || !up2_direct_batched(p, m1, ncls)) return CS_OK;
    CsConvGemm q = p;
    q.kd = p.ud ? 2 : 3; q.kh = p.uh ? 2 : 3; q.kw = p.uw ? 2 : 3;
    q.ud = q.uh = q.uw = 0;
    q.dout = p.din; q.hout = p.hin; q.wout = p.win;
    q.tile = 0;
    tile_dims(up2_batched_tile(p, m1, ncls), bm, bn);
    rps = (int64_t)p.din * p.hin * p.win;
    if (!bm || !p.bias) return CS_OK;                         // (piped epilogue needs bias / residual / row vector)
    // ... the rest of the kernel's `piped` condition: no BN scale / shift, and the rows one tile's scattered
    // stores span on the doubled grid -- at most the tile's rows plus a plane, a line and a voxel of halo, times the
    // doubling factor -- inside a 32-bit offset window
    if (p.scale) return CS_OK;
    const int64_t ospan = ((int64_t)bm + 2LL * p.hin * p.win + 2LL * p.win + 2) * ncls;
    if (ospan * p.ldo * 4 >= 0x7FF00000LL) return CS_OK;
    if (gn_rows && rps % bm == 0 && p.act != CS_ACT_GEGLU) *gn_rows = bm;
    return CS_OK;                                            // (no pair output from the scattered store)
  }
  if (p.a_format == 3 || p.a_format == 4) {      // the Winograd-W output-transform launch (wino_out_kernel): its workgroup's rows
    const int wr = p.a_format == 4 ? 64 : 32;
    if (gn_rows && rps % wr == 0 && p.act != CS_ACT_GEGLU) *gn_rows = wr;
    if (pair_ok && pair_geom && p.act != CS_ACT_GEGLU) *pair_ok = 1;
    return CS_OK;
  }
  if (p.splitk > 1) {
    if (gn_rows && rps % SKR == 0 && p.act != CS_ACT_GEGLU) *gn_rows = SKR;
    if (pair_ok && pair_geom && p.act != CS_ACT_GEGLU) *pair_ok = 1;
    return CS_OK;
  }
  int tile = p.tile ? p.tile : auto_tile(p, M, true);
  if (tile == 5) return CS_OK;                             // (removed tile code: no extras)
  tile_dims(tile, bm, bn);
  if (!bm) return CS_OK;
  if (tile == 10) {                        // the K-wave kernel's single epilogue emits both, whatever the epilogue terms
    if (!cs_kw_gemm_applicable(p, M)) return CS_OK;
    if (gn_rows && rps % bm == 0) *gn_rows = bm;
    if (pair_ok && pair_geom) *pair_ok = 1;
    return CS_OK;
  }
  // the kernel's `piped` condition, for every tile of the launch
  const bool piped = (p.res || p.bias || p.rowvec) && !p.scale && (p.act != CS_ACT_GEGLU || p.cout % bn == 0) &&
                     (!p.rowvec || p.rv_rows % bm == 0) &&
                     (int64_t)bm * p.ldo * 4 < 0x7FF00000LL && (!p.res || (int64_t)bm * p.ldr * 4 < 0x7FF00000LL);
  if (!piped) return CS_OK;
  if (gn_rows && rps % bm == 0 && p.act != CS_ACT_GEGLU) *gn_rows = bm;
  if (pair_ok && pair_geom) *pair_ok = 1;
  return CS_OK;
}

extern "C" int cs_conv_gemm_plan(const CsConvGemm* d, int32_t* splitk, int64_t* ws_bytes) {
  if (!d || !splitk) return CS_EINVAL;
  const int64_t M = (int64_t)d->nb * d->dout * d->hout * d->wout;
  if (M <= 0 || d->cout <= 0) return CS_EINVAL;
  const int s = plan_splitk(*d, M);
  *splitk = s;
  if (ws_bytes) *ws_bytes = s > 1 ? (int64_t)s * M * d->cout * 4 : 0;
  return CS_OK;
}

// most 64x64 tiles a launch may have and still take the K-wave kernel (one 128 KB workgroup per CU): four rounds of the chip
// (r5, 7 objects: 24.09 / 24.05 ms per step at one round, 23.94 / 23.92 at four -- the level-2 C x C GEMMs of 616 tiles ran
// at 73 TF/s on the one-chain tile, profiles/r05_e_kwave_tiles_ab.txt); CS_KWAVE_MAX_TILES overrides (tuning sweeps)
static int64_t kwave_max_tiles() {
  static const int64_t v = [] {
    const char* e = getenv("CS_KWAVE_MAX_TILES");
    return (e && *e) ? atoll(e) : (long long)-1;
  }();
  return v >= 0 ? v : 4LL * device_cus();
}

static int tok_t2_maxk() {
  static const int v = [] {
    const char* e = getenv("CS_TOK_T2_MAXK");
    return (e && *e) ? atoi(e) : 128;
  }();
  return v;
}

// tile of an automatic (desc->tile == 0) launch
static int auto_tile(const CsConvGemm& p, int M, bool f16x3) {
  int tile = 0;
  {
    // largest tile that still gives every one of the 256 CUs a workgroup; small problems take the 64x64 tile
    // (measured at CFG batch 2: 64x64 is ~2x the 128x224 tile, which leaves 3/4 of the chip idle)
    const int64_t mt = (M + 127) / 128;
    // F16X3 only: the 256x224 8-wave tile issues a third fewer LDS-DMA instructions per flop (measured +3..6 %
    // over 128x224) and wins as soon as it still occupies ~3/4 of the CUs
    if (f16x3 && p.cout % 224 == 0 && ((M + 255) / 256) * (int64_t)(p.cout / 224) >= 192)
      tile = 4;
    // the VQ-VAE decoder's channel counts (64 / 128 / 256) are not multiples of 224: 256-row tiles with 128 or 64
    // columns move a quarter to a third fewer LDS-DMA bytes per flop than the 128x128 / 64x64 tiles
    else if (f16x3 && p.cout % 128 == 0 && ((M + 255) / 256) * (int64_t)(p.cout / 128) >= 192)
      tile = 6;
    // ... and so do convs with a handful of output channels (the UNet's 224 -> 3, the decoder's 64 -> 1): the A
    // stream is what they pay for, and the 256-row tile moves it with fewer DMA instructions (583 vs 836 us and
    // 1304 vs 1744 us against the 64x64 tile, tools/small_n_bench.py)
    else if (f16x3 && p.cout <= 64 && (p.cout == 64 || p.cout <= 4) && (M + 255) / 256 >= 192)
      tile = 7;
    else if (p.cout % 224 == 0 && mt * (p.cout / 224) >= 256)
      tile = 2;
    else if (p.cout > 64 && mt * ((p.cout + 127) / 128) >= 256)
      tile = 1;
    else
      tile = 3;
    // (r3: tiles 8 / 9 -- 512 x 64 / 512 x 128, two row blocks per wave over one slab -- exist and are bit-identical, but
    // are NOT auto-selected: on the VQ-VAE decoder's 64^3 / 32^3 convs they measured 334-360 / 405-443 TF/s against
    // 366-377 / 434-451 for the 256-row tiles, profiles/r03_g_decode_tables.txt.  CS_TILE512=1 selects them for A/B runs.)
    {
      const bool t512on = cs_debug()->tile512 != 0;
      const bool conv3 = p.kd == 3 && p.kh == 3 && p.kw == 3 && p.sd == 1 && p.sh == 1 && p.sw == 1 &&
                         !(p.ud | p.uh | p.uw) && p.pd == 1 && p.ph == 1 && p.pw == 1;
      const int64_t t512 = (M + 511) / 512;
      if (t512on && conv3 && tile == 7 && p.cout == 64 && p.win <= 64 && t512 >= 512) tile = 8;
      if (t512on && conv3 && tile == 6 && p.win <= 32 && t512 * (p.cout / 128) >= 512) tile = 9;
    }
    // r4, 1-tap GEMMs (tools/gemm_tok_smallm.py): (i) the 256x224 tile runs one workgroup per CU, so a launch of r = tiles /
    // 256 rounds pays ceil(r): where the last round is mostly empty (ceil(r) / r >= 1.25: 672 -> 2016 at 16384 rows is 576
    // tiles = 2.25 rounds, 448 -> 1344 at 14336 rows 1.31) the 128x128 tile's finer grain wins -- 145 vs 179 us, 74 vs 94;
    // (ii) with a SHORT K loop (<= 32 chunks) and the operand pair already split, two 128-row workgroups per CU overlap
    // one's epilogue with the other's K loop, as in the fused-gate rule below: 448 -> 1344 at 65536 rows 281 vs 300 us.
    if (f16x3 && tile == 4 && p.kd * p.kh * p.kw == 1 && p.act != CS_ACT_GEGLU && !cs_debug()->no_tok_rules) {
      const int64_t t4 = ((M + 255) / 256) * (int64_t)(p.cout / 224);
      const int64_t rounds = (t4 + 255) / 256;
      if (t4 > 256 && rounds * 256 * 4 >= t4 * 5)
        tile = 1;
      else if (p.a_format == 2 && (p.cin + 15) / 16 <= 32)
        tile = 2;
      // (iii, r5) a single, not even full round of 256-row tiles and K <= 84 chunks (the level-0 skip convs at the reference's
      // mini-batch of 7: 224 tiles; the 672-column GEMMs at 16384 rows: 192) -- two 128-row workgroups per CU instead: 50.1
      // vs 58.6 us (448 -> 224) and 66.7 vs 72.8 (672 -> 224) at 57344 rows, 63.2 vs 68.0 (672 -> 672), 106.2 vs 111.7 (1344
      // -> 672) at 16384; 2688 -> 672 (168 chunks) stays: 193.8 vs 183.8.  profiles/r05_tok_smallm_b{14,64}.txt
      else if (t4 <= 256 && (p.cin + 15) / 16 <= 84)
        tile = 2;
      // (iv, r6) the same K bound for fp32 operands and for launches of several rounds: in the r5 sweep the 128-row pair of
      // workgroups wins every one-tap shape up to 84 chunks at batch 64 whatever the operand format (448 -> 448 f32 at 65536
      // rows 125.3 vs 130.9 us, 448 -> 224 at 262144 rows 220.0 vs 253.9, 672 -> 224 304.9 vs 322.8, 672 -> 448 140.3 vs
      // 151.8, 1120 -> 448 216.0 vs 226.1; at 112 chunks the tiles tie in isolation -- 1792 -> 448 319.9 vs 321.8 -- and the
      // 128-row pair wins inside the step; 2688 -> 672, 168 chunks, loses).  Same-box step A/B (profiles/r06_p_tok_t2_rule_*.txt):
      // bound 0 / 84 / 112 / 128 / 176 chunks = 64.26 / 63.86 resp. 63.25 / 63.00 / 62.94 / 63.01 ms per 32-object step -> 128.
      // CS_TOK_T2_MAXK overrides the bound (0 = rules (ii) / (iii) only: A/B runs).
      else if ((p.cin + 15) / 16 <= tok_t2_maxk())
        tile = 2;
    }
    // the fused gate needs whole [x | gate] 224-column tiles; two 128-row workgroups per CU overlap one's gate epilogue with
    // the other's K loop at BOTH widths (r5 sweeps, profiles/r05_tok_smallm_b{14,64}.txt: 672 -> 5376 at 3584 rows 99.0 vs
    // 113.5 us on the 256-row tile, at 16384 rows 363.8 vs 377.7; 448 -> 3584 at 65536 rows 717 vs 737)
    if (p.act == CS_ACT_GEGLU) tile = 2;
    // r5 (VERDICT r4 next #1a): small one-tap GEMMs -- every 64x64 tile resident at once (one 128 KB workgroup per CU) and
    // at least two K chunks per wave -- take the K-wave kernel: the four waves of a workgroup each run a quarter of the K
    // loop over the whole 64x64 tile, four accumulator chains and no barrier in the loop (cs_gemm_kw.hip).
    // CS_NO_KWAVE=1: the one-chain 64x64 tile (A/B runs).
    // (cout % 224 == 0: the UNet's channel counts.  The VQ decoder's 1x1x1 convs -- 64 / 128 / 256 / 768 columns -- keep the
    // one-chain tile at EVERY batch: the K-wave kernel partitions the K sum, and a decoder whose tile followed the batch
    // would lose its bit-exact batch invariance, tests/test_model_gpu.py::test_vq_decode_batch_invariance)
    if (f16x3 && tile == 3 && !cs_debug()->no_kwave && (p.cin + 15) / 16 >= 8 && p.cout % 224 == 0 &&
        cs_kw_gemm_applicable(p, M) && (int64_t)((M + 63) / 64) * ((p.cout + 63) / 64) <= kwave_max_tiles())
      tile = 10;
  }
  return tile;
}

// tile of a K-sliced launch (desc->splitk > 1): explicit narrow tiles (6 / 7) as given; the 256x224 tile for the large-batch
// four-way cut, an explicit tile 4 and -- r3 -- every 3x3x3 slab conv (CS_SLICE_TILE2=1: those on the 128-row tile); else 128x224
static int sliced_tile(const CsConvGemm& p, int M) {
  if (p.tile == 6 || p.tile == 7) return p.tile;
  const bool slab3 = !cs_debug()->slice_tile2 && p.kd == 3 && p.kh == 3 && p.kw == 3 && p.sd == 1 && p.sh == 1 &&
                     p.sw == 1 && p.pd == 1 && p.ph == 1 && p.pw == 1 && !(p.ud | p.uh | p.uw) && p.din == p.dout &&
                     p.hin == p.hout && p.win == p.wout && p.win <= 32;
  return (split4_large(p, M) || p.tile == 4 || slab3) ? 4 : 2;
}

// CUs of the current device (cached per process; 0 if the query fails -- then nothing is assumed resident)
static int device_cus() {
  static std::atomic<int> cached{-1};
  int v = cached.load(std::memory_order_relaxed);
  if (v >= 0) return v;
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
    n = 0;
  (void)hipGetLastError();
  cached.store(n, std::memory_order_relaxed);
  return n;
}

// Reducers per output tile of a K-sliced launch whose reduce + epilogue run inside the slice kernel (fused_splitk_reduce,
// cs_gemm_f16x3.hip), or 0: the two-kernel form.  The reducers WAIT for their tile's other slices, so every workgroup of
// the launch must be resident at once: one 8-wave workgroup per CU on the 256-row tiles, two 4-wave ones on the 128-row
// tile (launch bounds and LDS footprints of cs_gemm_f16x3.hip) -- the small-batch plan's launches are sized to exactly
// that; the large-batch four-way cut (768 workgroups) keeps the second launch.  Each reducer takes BM / R rows in whole
// 16-row statistics blocks: R = the largest power of two <= min(slices, BM / 16).
static int fused_reduce_plan(const CsConvGemm& p, int M, int stile) {
  if (!p.splitk_sync || cs_debug()->no_fused_reduce || p.splitk < 2 || (stile != 2 && stile != 4)) return 0;
  if (p.cout % 224 || ((uintptr_t)p.splitk_sync & 3)) return 0;
  const int bm = stile == 4 ? 256 : 128;
  const int64_t tiles = (int64_t)((M + bm - 1) / bm) * (p.cout / 224);
  const int64_t slots = (int64_t)device_cus() * (stile == 4 ? 1 : 2);
  if (tiles * p.splitk > slots || 2 * tiles > p.splitk_sync_words) return 0;
  if ((int64_t)p.splitk * M * p.cout * 4 >= 0xFFE00000LL) return 0;      // the reducers address the partials through one descriptor
  int r = 1;
  while (2 * r <= p.splitk && 2 * r <= bm / 16) r *= 2;
  return r;
}

// omap_f / omap_p != 0 (cs_conv_gemm_up2 only): this GEMM is one output parity class of a folded Upsample conv and
// stores straight into the doubled grid (cs_gemm_f16x3.hip, slab4 kernel); d->out / d->ldo are then the final tensor's
// ---------------------------------------------------------------------------------------------------------------------
// r5: Winograd F(2,3) along W for the 3x3x3 stride-1 convs (CsConvGemm.a_format = 3; every ResBlock conv,
// openai_model_3d.py:294-314).  Four position GEMMs with a 3x3x1 kernel over (D, H, W/2) -- 18 instead of 27 multiply-adds
// per output -- on the slab kernel's three-taps-per-kd instantiation (cs_gemm_f16x3.hip, TPK = 3), all four in ONE launch
// (operands stacked along the batch, four consecutive packed weight images), then the output transform in the split-K
// reduce + epilogue kernel's place.  Prototype on the per-tap gather path, separate launches: -21...-23 % GEMM time at the
// 16^3 / 16x8x8 levels, rel-L2 vs fp64 2.5e-7...3.3e-7 (direct form 2.6e-7...5.2e-7), profiles/r05_q_wino_proto.txt.
// ---------------------------------------------------------------------------------------------------------------------
// tile of the position GEMMs by output width: 256x224 (the UNet's widths), 256x128 / 256x64 (the VQ decoder's), 0 = none
static int wino_tile(int cout) { return cout % 224 == 0 ? 4 : cout % 128 == 0 ? 6 : cout == 64 ? 7 : 0; }

// the variant of the Winograd-W route this conv takes: 0 = none (direct form), 2 = F(2,3), 4 = F(4,3)
static int wino_variant(const CsConvGemm& p);
static bool wino_ok(const CsConvGemm& p) { return wino_variant(p) != 0; }
// F(2,3): the conditions of the route as such
static bool wino23_ok(const CsConvGemm& p) {
  const CsDebug* dbg = cs_debug();
  if (dbg->no_wino || p.math != CS_MATH_F16X3 || p.act == CS_ACT_GEGLU) return false;
  if (!(p.kd == 3 && p.kh == 3 && p.kw == 3 && p.sd == 1 && p.sh == 1 && p.sw == 1 && p.pd == 1 && p.ph == 1 && p.pw == 1) ||
      (p.ud | p.uh | p.uw) || p.din != p.dout || p.hin != p.hout || p.win != p.wout)
    return false;
  if ((p.win & 1) || p.win < 4 || p.win / 2 > 32 || !wino_tile(p.cout) || (p.cin & 7) || p.cin < 16) return false;
  // The VQ decoder's widths (64 / 128 / 256 columns) only at its 16^3 level: at 32^3 and 64^3 the transformed operand, the
  // position results and the output transform are passes over 1 - 8 GB per conv that cost more than a third of a K loop of
  // 36 - 144 chunks saves (decode of 32 objects 54.7 ms against 51.3 with every level on the route,
  // profiles/r05_x_decode_table.txt).  A rule on the SAMPLE's geometry: the decoder's route never follows the batch.
  if (p.cout % 224 && (int64_t)p.din * p.hin * p.win > 4096) return false;
  const int64_t M = (int64_t)p.nb * p.dout * p.hout * p.wout;
  const int64_t min_rows = dbg->wino_min_rows > 0 ? dbg->wino_min_rows : 1024;
  if (M < min_rows || M > 0x3fffffffLL || (M / 2) % 256) return false;
  // (32-bit offsets of the position GEMMs' slab windows and of the reduce kernel's workspace reads)
  if ((512 + 2LL * p.hin * (p.win / 2) + p.win + 32) * p.lda * 4 >= 0x7FF00000LL) return false;
  return true;
}
// F(4,3) (a_format = 4): six positions over M / 4 rows -- 13.5 of 27 multiply-adds, 1.5x (not 2x) operand / result passes, ~2x
// the direct form's rounding error (6e-7 - 8e-7 per conv, profiles/r05_z_wino43_numerics.txt): W % 4
// == 0, whole 256-row tiles per position, from CsDebug.wino43_min_rows rows
static int wino_variant(const CsConvGemm& p) {
  if (!wino23_ok(p)) return 0;
  const CsDebug* dbg = cs_debug();
  const int64_t M = (int64_t)p.nb * p.dout * p.hout * p.wout;
  const int64_t min43 = dbg->wino43_min_rows > 0 ? dbg->wino43_min_rows : 2048;
  // (the VQ decoder's widths too -- at its 16^3 level, the only one wino23_ok grants them: 4096 rows per sample, so the variant
  // does not follow the batch either while the threshold stays <= 4096)
  // (other widths -- the decoder's, the reduced test UNet's: the SAMPLE's rows decide, so that the variant cannot follow the batch)
  const int64_t rdec = p.cout % 224 ? (int64_t)p.din * p.hin * p.win : M;
  if (!dbg->no_wino43 && p.win % 4 == 0 && (M / 4) % 256 == 0 && (rdec / 4) % 256 == 0 && rdec >= min43) return 4;
  return 2;
}
// positions / outputs per tile of a variant
static inline int wino_pos(int variant) { return variant + 2; }

// K slices of the position GEMMs: their launch is 4 x (M / 512) x (cout / 224) workgroups of the 256x224 tile, one per CU.
// A small cost model in microseconds picks the slice count (whole super-chunks, padding <= 10 %, >= 2 super-chunks per
// slice): rounds of the chip x (a slice's K loop at ~0.95 us per 16-wide chunk + ~10 us of prologue / tile store) + the
// partial results each extra slice writes and the output transform reads back (~0.46 us per MB).  384 tiles at the 16x4x4
// level of 32 objects -> two slices (three even rounds, as the direct form's four-way cut); 448 tiles (the 16^3 level at
// 14 objects) -> one (the efficiency-only rule of the first version took four slices there and LOST 30 % to the direct
// form, profiles/r05_r_wino_bench_small.txt).
// r6: the plan of a position launch.  `slices` = the uniform slice count, or -- units_main > 0 -- the launch runs as TWO: the
// first units_main (position, column tile) units unsliced over whole rounds of the chip, then the remaining units cut into
// `slices` K slices that together fill (at most) one more round.  The workspace holds `slices` slices either way; the output
// transform reads one slice for the main units and `slices` for the tail's.  Why: 288 tiles (the 16x4x4 level at 32 objects:
// 6 positions x 3 column tiles = 18 units of 16 row tiles) on 256 CUs are 1.125 rounds -- three uniform slices ran as 3.4
// rounds of a third of the K loop each = 4 x 1/3 = 1.33 rounds' worth of time at 320 TF/s against 390-415 at the levels whose
// tile counts are whole rounds (DESIGN r5 open item (v)); 16 units = 256 workgroups for the whole K loop + 2 units x 8 slices =
// 256 workgroups for an eighth of it are 1 + 1/8.  The cut is by UNIT, not by row tile: every row of a unit is summed the same
// way, so a sample's result does not depend on where it sits in the batch (tests: objects with equal conditioning get equal
// shapes; a rank shard computed on its own equals the same rows of the whole launch).
struct WinoPlan {
  int slices;       // slices of the workspace (uniform plan: of every tile; tail plan: of the tail's tiles)
  int units_main;   // tail plan: (position, column tile) units of the unsliced main launch (0 = uniform plan)
};
static int wino_splits(const CsConvGemm& p, int variant, double* cost_us = nullptr);

static WinoPlan wino_plan(const CsConvGemm& p, int variant) {
  WinoPlan pl;
  double t_uni = 0;
  pl.slices = wino_splits(p, variant, &t_uni);
  pl.units_main = 0;
  static const bool forced = [] { const char* e = getenv("CS_WINO_SPLITS"); return e && *e && atoi(e) > 0; }();
  if (forced || cs_debug()->no_wino_tail || p.cout % 224) return pl;
  const int64_t M = (int64_t)p.nb * p.dout * p.hout * p.wout;
  const int64_t R = M / variant / 256, tiles_n = p.cout / 224;      // row tiles per position; column tiles
  const int64_t units = wino_pos(variant) * tiles_n;
  const int64_t cus = device_cus() > 0 ? device_cus() : 256;
  if (R < 1) return pl;
  const int64_t rounds = units * R / cus;                         // whole rounds the main launch may fill
  if (rounds < 1) return pl;
  const int64_t units_main = rounds * cus / R;                    // whole units
  const int64_t tail_tiles = (units - units_main) * R;
  if (units_main < 1 || tail_tiles < 1) return pl;
  const int64_t nsc = 3LL * ((p.cin + 15) / 16);
  int best = 0;
  for (int sp = 2; sp <= 16 && tail_tiles * sp <= cus; ++sp) {     // (one slice: the uniform plan's own extra round)
    const int64_t per = (nsc + sp - 1) / sp;
    if (per < 2 || per * sp * 10 > nsc * 11 || (nsc + per - 1) / per != sp) continue;
    best = sp;
  }
  if (!best) return pl;
  const int64_t per = (nsc + best - 1) / best;
  const double mb_tail = (double)tail_tiles * 256.0 * 224.0 * 4.0 / 1e6;
  double seam = (best - 1) * mb_tail * 0.46;
  if (seam < 2.0 * (best - 1)) seam = 2.0 * (best - 1);
  const double main_rounds = (double)((units_main * R + cus - 1) / cus);
  const double t_tail = main_rounds * ((double)nsc * 3.0 * 0.95 + 10.0) + ((double)per * 3.0 * 0.95 + 10.0) + seam + 5.0;
  if (t_tail < t_uni * 0.97) {
    pl.slices = best;
    pl.units_main = (int)units_main;
  }
  return pl;
}

static int wino_splits(const CsConvGemm& p, int variant, double* cost_us) {
  if (cost_us) *cost_us = 0.0;
  // (the VQ decoder's widths are never K-sliced: a slice count that followed the batch would cost the decoder its bit-exact
  // batch invariance, tests/test_model_gpu.py::test_vq_decode_batch_invariance)
  if (p.cout % 224) return 1;
  static const int forced = [] {              // CS_WINO_SPLITS: tuning sweeps only
    const char* e = getenv("CS_WINO_SPLITS");
    return (e && *e) ? atoi(e) : 0;
  }();
  const int64_t M = (int64_t)p.nb * p.dout * p.hout * p.wout;
  const int64_t tiles = wino_pos(variant) * (M / variant / 256) * (p.cout / 224);
  if (forced > 0) {
    const int64_t nsc0 = 3LL * ((p.cin + 15) / 16), per0 = (nsc0 + forced - 1) / forced;
    if (forced == 1 || (per0 >= 2 && per0 * forced * 10 <= nsc0 * 11 && (nsc0 + per0 - 1) / per0 == forced)) return forced;
  }
  const int64_t cus = device_cus() > 0 ? device_cus() : 256;
  const int64_t nsc = 3LL * ((p.cin + 15) / 16);
  const double mb = (double)wino_pos(variant) / variant * (double)M * p.cout * 4.0 / 1e6;      // one slice's position results
  int best = 1;
  double best_t = 0;
  for (int sp = 1; sp <= 16; ++sp) {
    const int64_t per = (nsc + sp - 1) / sp;
    if (sp > 1 && (per < 2 || per * sp * 10 > nsc * 11 || (nsc + per - 1) / per != sp)) continue;
    const int64_t rounds = (tiles * sp + cus - 1) / cus;
    double seam = (sp - 1) * mb * 0.46;
    if (sp > 1 && seam < 2.0 * (sp - 1)) seam = 2.0 * (sp - 1);
    const double t = (double)rounds * ((double)per * 3.0 * 0.95 + 10.0) + seam;
    if (sp == 1 || t < best_t * 0.97) {
      best = sp;
      best_t = t;
    }
  }
  if (cost_us) *cost_us = best_t;
  return best;
}

extern "C" int cs_conv_wino_ok(const CsConvGemm* d) { return (d && d->nb > 0 && d->cout > 0) ? wino_variant(*d) : 0; }

// the variant a descriptor asks for (a_format 3 / 4), 0 if this conv may not take it
static int wino_asked(const CsConvGemm& p) {
  const int v = p.a_format == 4 ? 4 : 2;
  const int best = wino_variant(p);
  return (best == 0 || v > best) ? 0 : v;          // (F(2,3) stays valid where F(4,3) is granted)
}

extern "C" int cs_conv_wino_plan(const CsConvGemm* d, int32_t* splitk, int64_t* ws_bytes) {
  if (!d || !splitk) return CS_EINVAL;
  const int v = wino_asked(*d);
  if (!v) return CS_EINVAL;
  const int sp = wino_plan(*d, v).slices;
  *splitk = sp;
  const int64_t M = (int64_t)d->nb * d->dout * d->hout * d->wout;
  if (ws_bytes) *ws_bytes = (int64_t)sp * wino_pos(v) * (M / v) * d->cout * 4;
  return CS_OK;
}

extern "C" int cs_conv_wino_plan_info(const CsConvGemm* d, int32_t* slices, int32_t* units_main, int32_t* units) {
  if (!d) return CS_EINVAL;
  const int v = wino_asked(*d);
  if (!v) return CS_EINVAL;
  const WinoPlan pl = wino_plan(*d, v);
  if (slices) *slices = pl.slices;
  if (units_main) *units_main = pl.units_main;
  if (units) *units = (int32_t)(wino_pos(v) * ((d->cout + 223) / 224));
  return CS_OK;
}

// desc validated by conv_gemm_impl (incl. gn_part / out_format against cs_conv_gemm_epilogue_caps)
// phases: bit 0 = the position GEMMs, bit 1 = output transform + epilogue (cs_conv_gemm: both)
static int conv_wino(const CsConvGemm& p, int M, hipStream_t s, int phases = 3) {
  auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
  const int v = wino_asked(p), np = wino_pos(v);
  if (!v || !p.x_lo || !al16(p.x_lo) || !p.w_lo || !al16(p.w_lo) || (p.lda & 7) || !p.splitk_ws || !al16(p.splitk_ws) ||
      p.a_bound || !(p.acc_scale > 0.f))
    return CS_EINVAL;
  if ((p.ldo & 3) || !al16(p.out) || (p.bias && !al16(p.bias)) || (p.scale && (!al16(p.scale) || !al16(p.shift))) ||
      (p.rowvec && ((p.ldrv & 3) || !al16(p.rowvec))) || (p.res && ((p.ldr & 3) || !al16(p.res))))
    return CS_EINVAL;
  const int sp = p.splitk > 1 ? p.splitk : 1;
  if (sp > 16) return CS_EINVAL;
  CsConvGemm q = p;
  q.nb = np * p.nb;
  q.win = q.wout = p.win / v;
  q.kw = 1;
  q.pw = 0;
  q.a_format = 1;
  q.out = reinterpret_cast<float*>(p.splitk_ws);
  q.ldo = p.cout;
  q.bias = q.scale = q.shift = q.rowvec = q.res = nullptr;
  q.act = CS_ACT_NONE;
  q.gn_part = nullptr;
  q.out_format = 0;
  q.splitk = 0;
  q.tile = wino_tile(p.cout);
  // r6: the tail plan (wino_plan) -- taken when the hosts pass the slice count cs_conv_wino_plan gave them; any other p.splitk
  // (tests, tuning sweeps) runs as that many uniform slices
  const WinoPlan pl = wino_plan(p, v);
  const bool tail = pl.units_main > 0 && pl.slices == sp;
  const int units = np * (p.cout / 224);
  if (phases & 1) {
    int rc;
    if (tail) {
      rc = cs_conv_gemm_f16x3_dispatch(q, np * (M / v), q.tile, 1, s, 16, np, nullptr, nullptr, nullptr, 0, nullptr, 0, pl.units_main);
      if (rc == CS_OK)
        rc = cs_conv_gemm_f16x3_dispatch(q, np * (M / v), q.tile, sp, s, 16, np, nullptr, nullptr, nullptr, 0, nullptr, pl.units_main,
                                         units - pl.units_main);
    } else {
      rc = cs_conv_gemm_f16x3_dispatch(q, np * (M / v), q.tile, sp, s, 16, np, nullptr, nullptr, nullptr, 0, nullptr);
    }
    if (rc != CS_OK) return rc;
  }
  if (!(phases & 2)) return CS_OK;
  const int tail_unit0 = tail ? pl.units_main : 0x7fffffff;
  const int sp_main = tail ? 1 : sp;
  // (one tile of v outputs per thread: 16 v rows x 64 columns per workgroup = the statistics tile cs_conv_gemm_epilogue_caps names)
  const int64_t nblk = (int64_t)((M / v + 15) / 16) * ((p.cout + 63) / 64);
  if (nblk > 0x7fffffffLL || (p.gn_part && p.gn_rows != 16 * v)) return CS_EINVAL;
  if (v == 4) {
    CS_LAUNCH(wino_out_kernel<4>, dim3((unsigned)nblk), dim3(256), 0, s, p, reinterpret_cast<const float*>(p.splitk_ws), M, sp_main,
              tail_unit0, sp, 224);
  } else {
    CS_LAUNCH(wino_out_kernel<2>, dim3((unsigned)nblk), dim3(256), 0, s, p, reinterpret_cast<const float*>(p.splitk_ws), M, sp_main,
              tail_unit0, sp, 224);
  }
  CS_CHECK_LAUNCH();
  return CS_OK;
}

static int conv_gemm_impl(const CsConvGemm* d, cs_stream_t stream, int omap_f, int omap_p,
                          const void* const* cls_w = nullptr, const void* const* cls_w_lo = nullptr,
                          const float* cls_acc = nullptr, int ncls = 0, int wino_phases = 3) {
  if (!d || !d->x || !d->w || !d->out) return CS_EINVAL;
  const CsConvGemm& p = *d;
  if (p.nb <= 0 || p.cin <= 0 || p.cout <= 0 || p.dout <= 0 || p.hout <= 0 || p.wout <= 0)
    return CS_EINVAL;
  const bool f16x3 = p.math == CS_MATH_F16X3;
  if ((p.cin & 3) || (p.lda & 3) || p.lda < p.cin) return CS_EINVAL;
  if (!f16x3 && ((p.ldw & 3) || p.ldw < p.cout))
    return CS_EINVAL;
  if (((uintptr_t)p.x & 15) || ((uintptr_t)p.w & 15)) return CS_EINVAL;
  if (p.kd <= 0 || p.kh <= 0 || p.kw <= 0 || p.sd <= 0 || p.sh <= 0 || p.sw <= 0) return CS_EINVAL;
  if (p.ud < 0 || p.uh < 0 || p.uw < 0 || p.ud > 4 || p.uh > 4 || p.uw > 4) return CS_EINVAL;
  if (p.scale && !p.shift) return CS_EINVAL;
  if (p.rowvec && p.rv_rows <= 0) return CS_EINVAL;
  const int out_cols = (p.act == CS_ACT_GEGLU) ? p.cout / 2 : p.cout;
  if (p.act == CS_ACT_GEGLU && (!f16x3 || (p.cout & 1))) return CS_EINVAL;
  if (p.ldo < out_cols || (p.res && p.ldr < p.cout)) return CS_EINVAL;
  if (p.math != CS_MATH_FP32 && !f16x3) return CS_EINVAL;
  const int64_t M64 = (int64_t)p.nb * p.dout * p.hout * p.wout;
  if (M64 > 0x7fffffffLL) return CS_EINVAL;
  const int M = (int)M64;
  int tile = p.tile;
  if (tile == 5) return CS_EINVAL;                         // (the ping-pong kernel's code: removed in r6)
  if (tile == 0) tile = auto_tile(p, M, f16x3);
  hipStream_t s = (hipStream_t)stream;
  if (p.splitk > 1 && omap_f) return CS_EINVAL;
  if (p.a_bound && (!f16x3 || p.a_format != 0 || ((uintptr_t)p.a_bound & 3))) return CS_EINVAL;
  if ((p.gn_part || p.out_format) && !omap_f) {
    // epilogue outputs beside the fp32 tile: only where cs_conv_gemm_epilogue_caps says the launch can produce them
    int32_t rows = 0, pair = 0;
    if (!f16x3 || (p.out_format != 0 && p.out_format != 2) || cs_conv_gemm_epilogue_caps(d, &rows, &pair) != CS_OK)
      return CS_EINVAL;
    if (p.gn_part && (rows == 0 || rows != p.gn_rows || p.gn_ld < p.cout || ((uintptr_t)p.gn_part & 15))) return CS_EINVAL;
    if (p.out_format == 2 && (!pair || !(p.out_scale > 0.f))) return CS_EINVAL;
  }
  if (p.a_format == 3 || p.a_format == 4) return (f16x3 && !omap_f) ? conv_wino(p, M, s, wino_phases) : CS_EINVAL;
  if (wino_phases != 3) return CS_EINVAL;
  if (p.splitk > 1) {
    // caller-requested split-K (cs_conv_gemm_plan): partial tiles to the workspace, then reduce + epilogue
    auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    // (explicit tiles 6 / 7 -- 256x128 / 256x64, any cout % 4 == 0 -- are accepted for K-sliced launches too: tuning sweeps)
    const bool narrow = p.tile == 6 || p.tile == 7;
    if (!f16x3 || p.act == CS_ACT_GEGLU || (narrow ? (p.cout & 3) : p.cout % 224) || !p.splitk_ws || !al16(p.splitk_ws) ||
        p.splitk > 64)
      return CS_EINVAL;
    if ((p.ldo & 3) || !al16(p.out) || (p.bias && !al16(p.bias)) || (p.scale && (!al16(p.scale) || !al16(p.shift))) ||
        (p.rowvec && ((p.ldrv & 3) || !al16(p.rowvec))) || (p.res && ((p.ldr & 3) || !al16(p.res))))
      return CS_EINVAL;
    const int64_t nk = (int64_t)p.kd * p.kh * p.kw * ((p.cin + 15) / 16);
    if (p.splitk > nk) return CS_EINVAL;
    CsConvGemm part = p;
    part.out = reinterpret_cast<float*>(p.splitk_ws);
    part.ldo = p.cout;
    part.bias = part.scale = part.shift = part.rowvec = part.res = nullptr;
    part.act = CS_ACT_NONE;
    part.gn_part = nullptr;              // the slices write plain partial tiles; the reduce kernel emits the extras
    part.out_format = 0;
    // small batches: 128x224 tiles; the large-batch four-way cut (plan_splitk) keeps the 256x224 slab kernel
    // r3: K-sliced 3x3x3 slab convs run the 256x224 tile at EVERY batch size (one 8-wave workgroup per CU: 256 slots,
    // the same slice count as 512 slots of the 128-row tile) -- 4-7 % faster than the 128x224 tile on every shape from
    // 1 to 14 objects (tools/gemm_smallm_t4.py, profiles/r03_an_tile4_slices*.txt); convs the slab kernel does not take
    // (strided, W > 32) keep the 128-row tile.  CS_SLICE_TILE2=1: the previous rule, A/B runs.
    const int stile = sliced_tile(p, M);
    // r5: reduce + epilogue inside the slice kernel (CsConvGemm.splitk_sync) -- when every workgroup of the launch is resident
    // at once (the reducers wait for their tile's other slices), for whole 224-column tiles with the float4 epilogue
    CsFuseK fz;
    memset(&fz, 0, sizeof(fz));
    const int reducers = fused_reduce_plan(p, M, stile);
    if (reducers > 0) {
      fz.out = p.out; fz.bias = p.bias; fz.scale = p.scale; fz.shift = p.shift; fz.rowvec = p.rowvec; fz.res = p.res;
      fz.gn_part = p.gn_part; fz.sync = p.splitk_sync; fz.status = p.status;
      fz.ldo = p.ldo; fz.ldr = p.ldr; fz.ldrv = p.ldrv; fz.rv_rows = p.rv_rows; fz.act = p.act; fz.gn_ld = p.gn_ld;
      fz.out_format = p.out_format; fz.reducers = reducers; fz.out_scale = p.out_scale;
      return cs_conv_gemm_f16x3_dispatch(part, M, stile, p.splitk, s, 0, 0, nullptr, nullptr, nullptr, 0, &fz);
    }
    const int rc = cs_conv_gemm_f16x3_dispatch(part, M, stile, p.splitk, s);
    if (rc != CS_OK) return rc;
    if (p.gn_part || p.out_format) {
      const int64_t nblk = (int64_t)((M + SKR - 1) / SKR) * ((p.cout + 63) / 64);
      if (nblk > 0x7fffffffLL) return CS_EINVAL;
      CS_LAUNCH(splitk_reduce_epi_kernel, dim3((unsigned)nblk), dim3(256), 0, s, p,
                reinterpret_cast<const float*>(p.splitk_ws), M, p.splitk);
    } else {
      CS_LAUNCH(splitk_reduce_kernel, dim3(cs_grid_for((int64_t)M * (p.cout >> 2), 256)), dim3(256), 0, s, p,
                reinterpret_cast<const float*>(p.splitk_ws), M, p.splitk);
    }
    CS_CHECK_LAUNCH();
    return CS_OK;
  }
  if (omap_f && !f16x3) return CS_EINVAL;
  if ((p.gn_part || p.out_format) && !f16x3) return CS_EINVAL;
  if (tile == 10) return (f16x3 && !omap_f) ? cs_kw_gemm_f16x3_launch(p, M, s) : CS_EINVAL;
  if (f16x3) return cs_conv_gemm_f16x3_dispatch(p, M, tile, 1, s, omap_f, omap_p, cls_w, cls_w_lo, cls_acc, ncls);
  switch (tile) {
    case 1: return launch<2, 2, 2, 2>(p, M, s);
    case 2: return launch<1, 7, 4, 1>(p, M, s);
    case 3: return launch<1, 1, 2, 2>(p, M, s);
    case 4: return launch<1, 7, 4, 1>(p, M, s);   // the 256-row tiles exist for F16X3 only; same N tiling here
    case 6: return launch<2, 2, 2, 2>(p, M, s);
    case 7: return launch<1, 1, 2, 2>(p, M, s);
    case 8: return launch<1, 1, 2, 2>(p, M, s);
    case 9: return launch<2, 2, 2, 2>(p, M, s);
    default: return CS_EINVAL;
  }
}

extern "C" int cs_conv_gemm(const CsConvGemm* d, cs_stream_t stream) { return conv_gemm_impl(d, stream, 0, 0); }
extern "C" int cs_conv_wino_positions(const CsConvGemm* d, cs_stream_t stream) {
  return conv_gemm_impl(d, stream, 0, 0, nullptr, nullptr, nullptr, 0, 1);
}
extern "C" int cs_conv_wino_output(const CsConvGemm* d, cs_stream_t stream) {
  return conv_gemm_impl(d, stream, 0, 0, nullptr, nullptr, nullptr, 0, 2);
}

int cs_f16x3_slab_width(const CsConvGemm& p, int tile, int splits);      // cs_gemm_f16x3.hip

// the tile / slab choice conv_gemm_impl + cs_conv_gemm_f16x3_dispatch make for this descriptor, for hosts that account
// per kernel variant (bench.py) -- computed by the same functions the launch path calls
extern "C" int cs_conv_gemm_launch_info(const CsConvGemm* d, int32_t* tile_out, int32_t* slab_out) {
  if (!d || d->nb <= 0 || d->cout <= 0 || d->dout <= 0 || d->hout <= 0 || d->wout <= 0) return CS_EINVAL;
  const CsConvGemm& p = *d;
  const int64_t M64 = (int64_t)p.nb * p.dout * p.hout * p.wout;
  if (M64 > 0x7fffffffLL) return CS_EINVAL;
  const int M = (int)M64;
  const bool f16x3 = p.math == CS_MATH_F16X3;
  int tile = p.tile;
  if (p.a_format == 3 || p.a_format == 4) {   // the Winograd-W position GEMMs: 256-row tile, three-tap slab
    if (tile_out) *tile_out = wino_tile(p.cout);
    if (slab_out) *slab_out = 32;
    return CS_OK;
  }
  if (p.splitk > 1) {
    tile = sliced_tile(p, M);
  } else {
    if (tile == 5) return CS_EINVAL;
    if (tile == 0) tile = auto_tile(p, M, f16x3);
  }
  if (tile_out) *tile_out = tile;
  if (slab_out) *slab_out = f16x3 ? cs_f16x3_slab_width(p, tile, p.splitk > 1 ? p.splitk : 1) : 0;
  return CS_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Nearest x2 upsampling followed by a 3x3x3 stride-1 conv (Upsample: openai_model_3d.py:150-153, vqvae_modules.py:
// 35-39), computed on the SOURCE grid.  An output voxel 2h'+ph reads upsampled rows 2h'+ph-1 .. 2h'+ph+1, i.e. source
// rows {h'-1, h', h'} (ph = 0) or {h', h', h'+1} (ph = 1): per output parity the three taps of an upsampled dim
// collapse to TWO source taps with pre-summed weights ([w0, w1+w2] at offsets {-1, 0}; [w0+w1, w2] at {0, +1}), with the
// same zero padding.  So the op is 4 (H, W doubled) or 8 (D too) ordinary convs with 3x2x2 / 2x2x2 kernels whose
// outputs interleave: 12/27 resp. 8/27 of the multiply-adds of the direct form.  Sums of two fp32 weights round once
// (2^-24 relative), the same order as the fp32 accumulation they replace.
// ---------------------------------------------------------------------------------------------------------
namespace {

// w [cout*cin][3][3][3]  ->  wf [cls][cout*cin][kd'][kh'][kw'];  cls = (pd * nh + ph) * nw + pw over the doubled dims
__global__ __launch_bounds__(256) void fold_upsample_kernel(const float* __restrict__ w, float* __restrict__ wf,
                                                            int64_t cc, int ud, int uh, int uw) {
  const int kd = ud ? 2 : 3, kh = uh ? 2 : 3, kw = uw ? 2 : 3;
  const int nh = uh ? 2 : 1, nw = uw ? 2 : 1;
  const int taps = kd * kh * kw;
  const int ncls = (ud ? 2 : 1) * nh * nw;
  const int64_t total = (int64_t)ncls * cc * taps;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int t = (int)(i % taps);
    int64_t r = i / taps;
    const int64_t oc = r % cc;
    const int cls = (int)(r / cc);
    const int c = t % kw;
    t /= kw;
    const int b = t % kh;
    const int a = t / kh;
    const int pw = cls % nw, ph = (cls / nw) % nh, pd = cls / (nw * nh);
    // source taps of folded tap k at parity p: doubled dim -> (p == 0 ? k == 0 ? {0} : {1, 2} : k == 0 ? {0, 1} : {2})
    auto lo = [](int u, int p, int k) { return !u ? k : (p == 0 ? (k == 0 ? 0 : 1) : (k == 0 ? 0 : 2)); };
    auto hi = [](int u, int p, int k) { return !u ? k : (p == 0 ? (k == 0 ? 0 : 2) : (k == 0 ? 1 : 2)); };
    float acc = 0.f;
    for (int x = lo(ud, pd, a); x <= hi(ud, pd, a); ++x)
      for (int y = lo(uh, ph, b); y <= hi(uh, ph, b); ++y)
        for (int z = lo(uw, pw, c); z <= hi(uw, pw, c); ++z) acc += w[oc * 27 + (x * 3 + y) * 3 + z];
    wf[i] = acc;
  }
}

// tmp [cls][M1][cout] (source-grid outputs per parity class)  ->  out rows of the doubled grid
__global__ __launch_bounds__(256) void up2_interleave_kernel(const float* __restrict__ tmp, float* __restrict__ out,
                                                             int64_t m1, int cout, int ldo, int D, int H, int W, int ud,
                                                             int uh, int uw) {
  const int nh = uh ? 2 : 1, nw = uw ? 2 : 1;
  const int cls = blockIdx.y;
  const int pw = cls % nw, ph = (cls / nw) % nh, pd = cls / (nw * nh);
  const int c4n = cout >> 2;
  const int64_t total = m1 * c4n;
  const float* src = tmp + (int64_t)cls * m1 * cout;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % c4n);
    int64_t m = i / c4n;
    const int64_t mm = m;
    const int w_ = (int)(m % W);
    m /= W;
    const int h_ = (int)(m % H);
    m /= H;
    const int d_ = (int)(m % D);
    const int64_t n = m / D;
    const int64_t orow = ((n * (D << ud) + ((d_ << ud) + pd)) * (H << uh) + ((h_ << uh) + ph)) * (W << uw) + ((w_ << uw) + pw);
    *reinterpret_cast<f32x4*>(out + orow * ldo + c4 * 4) = *reinterpret_cast<const f32x4*>(src + mm * cout + c4 * 4);
  }
}

// r5: the K-sliced class batch of small launches (cs_conv_gemm_up2): ws [cls][slice][M1][cout] partial tiles -> the sum over
// the slices IN SLICE ORDER + bias + activation, stored to the class's rows of the doubled grid.  One launch instead of a
// reduce per class and the interleave pass; per element the arithmetic of splitk_reduce_kernel.
__global__ __launch_bounds__(256) void up2_reduce_scatter_kernel(const float* __restrict__ ws, const float* __restrict__ bias,
                                                                 float* __restrict__ out, int64_t m1, int cout, int ldo,
                                                                 int D, int H, int W, int ud, int uh, int uw, int splits,
                                                                 int act) {
  const int nh = uh ? 2 : 1, nw = uw ? 2 : 1;
  const int cls = blockIdx.y;
  const int pw = cls % nw, ph = (cls / nw) % nh, pd = cls / (nw * nh);
  const int c4n = cout >> 2;
  const int64_t total = m1 * c4n;
  const int64_t sstride = m1 * cout;
  const float* src = ws + (int64_t)cls * splits * sstride;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % c4n);
    int64_t m = i / c4n;
    const float* sp = src + m * cout + c4 * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(sp);
    for (int s0 = 1; s0 < splits; s0 += 8) {
      f32x4 t[8];
#pragma unroll
      for (int q = 0; q < 8; ++q)
        if (s0 + q < splits) t[q] = *reinterpret_cast<const f32x4*>(sp + (s0 + q) * sstride);
#pragma unroll
      for (int q = 0; q < 8; ++q)
        if (s0 + q < splits) v += t[q];
    }
    if (bias) v += *reinterpret_cast<const f32x4*>(bias + c4 * 4);
    if (act != CS_ACT_NONE) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = cs_act(v[e], act);
    }
    const int w_ = (int)(m % W);
    m /= W;
    const int h_ = (int)(m % H);
    m /= H;
    const int d_ = (int)(m % D);
    const int64_t n = m / D;
    const int64_t orow = ((n * (D << ud) + ((d_ << ud) + pd)) * (H << uh) + ((h_ << uh) + ph)) * (W << uw) + ((w_ << uw) + pw);
    *reinterpret_cast<f32x4*>(out + orow * ldo + c4 * 4) = v;
  }
}

bool up2_ok(const CsConvGemm& p) {
  return p.kd == 3 && p.kh == 3 && p.kw == 3 && p.sd == 1 && p.sh == 1 && p.sw == 1 && p.pd == 1 && p.ph == 1 &&
         p.pw == 1 && (p.ud | p.uh | p.uw) != 0 && p.ud >= 0 && p.ud <= 1 && p.uh >= 0 && p.uh <= 1 && p.uw >= 0 &&
         p.uw <= 1 && p.dout == (p.din << p.ud) && p.hout == (p.hin << p.uh) && p.wout == (p.win << p.uw) &&
         !p.res && !p.rowvec && !p.scale && p.act != CS_ACT_GEGLU && (p.cout & 3) == 0 && (p.ldo & 3) == 0 &&
         p.splitk <= 1;
}

}  // namespace

// Will cs_conv_gemm_up2 run this folded Upsample conv as ONE launch whose parity classes store straight into the doubled
// grid?  (Every class on the four-tap slab kernel, unsplit; CS_NO_UP2_DIRECT / CS_NO_UP2_BATCH switch the route off.)
static bool up2_direct_batched(const CsConvGemm& d, int64_t m1, int ncls) {
  if (d.math != CS_MATH_F16X3 || !up2_ok(d) || m1 > 0x7fffffffLL) return false;
  if (cs_debug()->no_up2_direct || cs_debug()->no_up2_batch || m1 * (int64_t)ncls > 0x7fffffffLL) return false;
  const int nh = d.uh ? 2 : 1, nw = d.uw ? 2 : 1;
  // r5: medium batches (seven objects' 4^3 -> 8x8 Upsample conv: four classes of 14 x 3 tiles) -- a class ALONE has few
  // enough tiles for the plan to slice it (four K-sliced launches + four reduces + the interleave: 1.5 ms per step at 7
  // objects), but all classes TOGETHER are 168 workgroups: one unsliced launch with the scattered store, from half the chip up
  const int bn_all = d.cout % 224 == 0 ? 224 : 128;
  const bool fills = d.cout % bn_all == 0 && (int64_t)ncls * ((m1 + 255) / 256) * (d.cout / bn_all) >= 128;
  for (int cls = 0; cls < ncls; ++cls) {
    const int pw = cls % nw, ph = (cls / nw) % nh, pd = cls / (nw * nh);
    CsConvGemm q = d;
    q.kd = d.ud ? 2 : 3; q.kh = d.uh ? 2 : 3; q.kw = d.uw ? 2 : 3;
    q.pd = d.ud ? 1 - pd : 1; q.ph = d.uh ? 1 - ph : 1; q.pw = d.uw ? 1 - pw : 1;
    q.ud = q.uh = q.uw = 0;
    q.dout = d.din; q.hout = d.hin; q.wout = d.win;
    q.ldo = d.cout;
    q.tile = 0;
    q.splitk = 0;
    const int tl = fills ? (bn_all == 224 ? 4 : 6) : auto_tile(q, (int)m1, true);
    if ((!fills && plan_splitk(q, m1) > 1) || !cs_f16x3_slab4_ok(q, tl, 1)) return false;
  }
  if (fills && (m1 > 0x7fffffffLL)) return false;
  return (int64_t)ncls * ((m1 + 255) / 256) * ((d.cout + 63) / 64) < 0x7fffffffLL;
}

// r5: K slices of the ONE-launch class batch for small folded-Upsample launches, or 0.  One object's 4^3 -> 8x8 Upsample
// conv is four 512-row GEMMs (672 -> 672, twelve taps): run per class they are 12 output tiles each -- 16 K slices and a
// reduce per class plus the interleave pass, nine launches, ~260 us for 22 GFLOP.  All classes' tiles x S slices fill
// the chip once instead, on the four-tap slab kernel (256-row tiles), and ONE reduce-scatter writes the doubled grid.
static int up2_sliced_plan(const CsConvGemm& d, int64_t m1, int ncls) {
  if (d.math != CS_MATH_F16X3 || !up2_ok(d) || d.a_format != 0 || d.gn_part || d.out_format || m1 > 0x7fffffffLL) return 0;
  if (cs_debug()->no_up2_direct || cs_debug()->no_up2_batch || cs_debug()->no_slab4) return 0;
  const int tile = d.cout % 224 == 0 ? 4 : d.cout % 128 == 0 ? 6 : 0;
  if (!tile) return 0;
  const int bn = tile == 4 ? 224 : 128;
  CsConvGemm q = d;
  q.kd = d.ud ? 2 : 3; q.kh = d.uh ? 2 : 3; q.kw = d.uw ? 2 : 3;
  q.pd = q.ph = q.pw = 1;
  q.ud = q.uh = q.uw = 0;
  q.dout = d.din; q.hout = d.hin; q.wout = d.win;
  q.tile = 0; q.splitk = 0;
  if (!cs_f16x3_slab4_ok(q, tile, -1)) return 0;
  const int64_t tiles = (int64_t)ncls * ((m1 + 255) / 256) * (d.cout / bn);
  const int64_t slots = device_cus();
  if (tiles * 2 > slots) return 0;                          // at least two slices: larger launches keep their routes
  const int64_t nsc = (int64_t)q.kd * ((d.cin + 15) / 16);  // super-chunks (one kd x 16 channels = four taps)
  int64_t sl = slots / tiles;
  if (sl > 32) sl = 32;
  if (sl > nsc / 2) sl = nsc / 2;
  if (sl < 2) return 0;
  sl = (nsc + (nsc + sl - 1) / sl - 1) / ((nsc + sl - 1) / sl);      // the fewest slices of that length
  if ((int64_t)ncls * sl * m1 * d.cout * 4 >= 0x7FF00000LL * 4) return 0;
  return sl < 2 ? 0 : (int)sl;
}

// tile of the one-launch batched route (0: the route is not taken): the automatic per-class tile, or -- where only all
// classes together fill half the chip (up2_direct_batched) -- the 256-row tile of the channel count
static int up2_batched_tile(const CsConvGemm& d, int64_t m1, int ncls) {
  if (!up2_direct_batched(d, m1, ncls)) return 0;
  CsConvGemm q = d;
  q.kd = d.ud ? 2 : 3; q.kh = d.uh ? 2 : 3; q.kw = d.uw ? 2 : 3;
  q.ud = q.uh = q.uw = 0;
  q.dout = d.din; q.hout = d.hin; q.wout = d.win;
  q.tile = 0; q.splitk = 0;
  const int t = auto_tile(q, (int)m1, true);
  if (t == 4 || t == 6) return t;
  return d.cout % 224 == 0 ? 4 : 6;
}

extern "C" int cs_conv_up2_info(int ud, int uh, int uw, int32_t* ncls, int32_t* kd, int32_t* kh, int32_t* kw) {
  if (ud < 0 || ud > 1 || uh < 0 || uh > 1 || uw < 0 || uw > 1 || !(ud | uh | uw)) return CS_EINVAL;
  if (ncls) *ncls = (ud ? 2 : 1) * (uh ? 2 : 1) * (uw ? 2 : 1);
  if (kd) *kd = ud ? 2 : 3;
  if (kh) *kh = uh ? 2 : 3;
  if (kw) *kw = uw ? 2 : 3;
  return CS_OK;
}

extern "C" int cs_fold_upsample_weight(const float* w_torch, float* w_folded, int cout, int cin, int ud, int uh, int uw,
                                       cs_stream_t stream) {
  int32_t ncls, kd, kh, kw;
  if (!w_torch || !w_folded || cout <= 0 || cin <= 0 || cs_conv_up2_info(ud, uh, uw, &ncls, &kd, &kh, &kw)) return CS_EINVAL;
  const int64_t cc = (int64_t)cout * cin;
  CS_LAUNCH(fold_upsample_kernel, dim3(cs_grid_for((int64_t)ncls * cc * kd * kh * kw, 256, 256 * 32)), dim3(256), 0,
            (hipStream_t)stream, w_torch, w_folded, cc, ud, uh, uw);
  CS_CHECK_LAUNCH();
  return CS_OK;
}

extern "C" int64_t cs_conv_gemm_up2_ws_bytes(const CsConvGemm* d) {
  if (!d || !up2_ok(*d)) return -1;
  const int64_t m1 = (int64_t)d->nb * d->din * d->hin * d->win;
  const int ncls = (d->ud ? 2 : 1) * (d->uh ? 2 : 1) * (d->uw ? 2 : 1);
  CsConvGemm q = *d;
  q.kd = d->ud ? 2 : 3; q.kh = d->uh ? 2 : 3; q.kw = d->uw ? 2 : 3;
  q.ud = q.uh = q.uw = 0;
  q.dout = d->din; q.hout = d->hin; q.wout = d->win;
  q.tile = 0;
  const int s = plan_splitk(q, m1);
  const int64_t per_class = ((int64_t)ncls + (s > 1 ? s : 0)) * m1 * d->cout * 4;
  const int sb = up2_sliced_plan(*d, m1, ncls);            // r5: [class][slice] partial tiles of the one-launch class batch
  const int64_t batched = (int64_t)ncls * sb * m1 * d->cout * 4;
  return per_class > batched ? per_class : batched;
}

extern "C" int cs_conv_gemm_up2(const CsConvGemm* d, const void* const* w_cls, const void* const* w_lo_cls,
                                const float* acc_scale_cls, void* ws, cs_stream_t stream) {
  if (!d || !w_cls || !ws || !d->x || !d->out || !up2_ok(*d) || ((uintptr_t)ws & 15) || ((uintptr_t)d->out & 15))
    return CS_EINVAL;
  const bool f16x3 = d->math == CS_MATH_F16X3;
  if (f16x3 && (!w_lo_cls || !acc_scale_cls)) return CS_EINVAL;
  const int64_t m1 = (int64_t)d->nb * d->din * d->hin * d->win;
  if (m1 > 0x7fffffffLL) return CS_EINVAL;
  const int nh = d->uh ? 2 : 1, nw = d->uw ? 2 : 1;
  const int ncls = (d->ud ? 2 : 1) * nh * nw;
  float* tmp = reinterpret_cast<float*>(ws);
  auto class_desc = [&](int cls) {
    const int pw = cls % nw, ph = (cls / nw) % nh, pd = cls / (nw * nh);
    CsConvGemm q = *d;
    q.kd = d->ud ? 2 : 3; q.kh = d->uh ? 2 : 3; q.kw = d->uw ? 2 : 3;
    q.pd = d->ud ? 1 - pd : 1; q.ph = d->uh ? 1 - ph : 1; q.pw = d->uw ? 1 - pw : 1;
    q.ud = q.uh = q.uw = 0;
    q.dout = d->din; q.hout = d->hin; q.wout = d->win;
    q.w = reinterpret_cast<const float*>(w_cls[cls]);
    q.w_lo = f16x3 ? w_lo_cls[cls] : nullptr;
    if (f16x3) q.acc_scale = acc_scale_cls[cls];
    q.out = tmp + (int64_t)cls * m1 * d->cout;
    q.ldo = d->cout;
    q.tile = 0;
    return q;
  };
  // r3: where every class runs the four-tap slab kernel unsplit, the classes store straight into the doubled grid (the
  // kernel's scattered-store epilogue) -- no scratch tensor, no interleave pass (2.06 of 57.8 ms per 32-object decode,
  // 0.23 ms of a UNet step).  CS_NO_UP2_DIRECT=1: scratch + interleave everywhere (A/B runs, the equality test).
  bool direct = f16x3 && !cs_debug()->no_up2_direct && m1 * (int64_t)ncls <= 0x7fffffffLL;
  for (int cls = 0; cls < ncls && direct; ++cls) {
    const CsConvGemm q = class_desc(cls);
    direct = plan_splitk(q, m1) <= 1 && cs_f16x3_slab4_ok(q, auto_tile(q, (int)m1, true), 1);
  }
  // ... and as ONE launch over all classes (virtual tile range [class][tile]): 192-workgroup class GEMMs (the UNet's 4^3
  // level at 32 objects) fill the chip together, and seven launch ramps go.  CS_NO_UP2_BATCH=1: one launch per class.
  const bool batched = up2_direct_batched(*d, m1, ncls);
  if (d->gn_part) {      // GroupNorm partials: the one-launch route only (tiles ordered [class][source-row tile])
    int32_t rows = 0;
    if (!batched || cs_conv_gemm_epilogue_caps(d, &rows, nullptr) != CS_OK || rows == 0 || rows != d->gn_rows ||
        d->gn_ld < d->cout || ((uintptr_t)d->gn_part & 15))
      return CS_EINVAL;
  }
  if (d->out_format) return CS_EINVAL;
  const int sliced = batched ? 0 : up2_sliced_plan(*d, m1, ncls);
  if (sliced > 1) {
    const int omap_f = (d->ud ? 4 : 0) | (d->uh ? 2 : 0) | (d->uw ? 1 : 0) | 8;
    CsConvGemm q = class_desc(0);
    q.out = tmp;
    q.ldo = d->cout;
    q.bias = nullptr;
    q.act = CS_ACT_NONE;
    q.gn_part = nullptr;
    const int tile = d->cout % 224 == 0 ? 4 : 6;
    const int rc = cs_conv_gemm_f16x3_dispatch(q, (int)m1, tile, sliced, (hipStream_t)stream, omap_f, 0, w_cls, w_lo_cls,
                                               acc_scale_cls, ncls);
    if (rc != CS_OK) return rc;
    const int64_t units = m1 * (d->cout >> 2);
    CS_LAUNCH(up2_reduce_scatter_kernel, dim3(cs_grid_for(units, 256, 256 * 8), ncls), dim3(256), 0, (hipStream_t)stream, tmp,
              d->bias, d->out, m1, d->cout, d->ldo, d->din, d->hin, d->win, d->ud, d->uh, d->uw, sliced, d->act);
    CS_CHECK_LAUNCH();
    return CS_OK;
  }
  if (batched) {
    const int omap_f = (d->ud ? 4 : 0) | (d->uh ? 2 : 0) | (d->uw ? 1 : 0);
    CsConvGemm q = class_desc(0);
    q.out = d->out;
    q.ldo = d->ldo;
    q.tile = up2_batched_tile(*d, m1, ncls);           // (explicit: a class alone might pick a smaller tile)
    return conv_gemm_impl(&q, stream, omap_f, 0, w_cls, w_lo_cls, acc_scale_cls, ncls);
  }
  if (direct) {
    const int omap_f = (d->ud ? 4 : 0) | (d->uh ? 2 : 0) | (d->uw ? 1 : 0);
    for (int cls = 0; cls < ncls; ++cls) {
      const int pw = cls % nw, ph = (cls / nw) % nh, pd = cls / (nw * nh);
      CsConvGemm q = class_desc(cls);
      q.out = d->out;
      q.ldo = d->ldo;
      const int rc = conv_gemm_impl(&q, stream, omap_f, (pd << 2) | (ph << 1) | pw);
      if (rc != CS_OK) return rc;
    }
    return CS_OK;
  }
  for (int cls = 0; cls < ncls; ++cls) {
    CsConvGemm q = class_desc(cls);
    const int s = plan_splitk(q, m1);
    q.splitk = s > 1 ? s : 0;
    q.splitk_ws = s > 1 ? (void*)(tmp + (int64_t)ncls * m1 * d->cout) : nullptr;
    const int rc = cs_conv_gemm(&q, stream);
    if (rc != CS_OK) return rc;
  }
  const int64_t units = m1 * (d->cout >> 2);
  CS_LAUNCH(up2_interleave_kernel, dim3(cs_grid_for(units, 256, 256 * 8), ncls), dim3(256), 0, (hipStream_t)stream, tmp,
            d->out, m1, d->cout, d->ldo, d->din, d->hin, d->win, d->ud, d->uh, d->uw);
  CS_CHECK_LAUNCH();
  return CS_OK;
}

static void fill_conv(CsConvGemm& p, const float* x, const float* w, const float* bias, float* out,
                      int nb, int d, int h, int w_, int cin, int cout, int sh, int sw) {
  p = CsConvGemm{};
  p.x = x; p.w = w; p.bias = bias; p.out = out;
  p.nb = nb; p.din = d; p.hin = h; p.win = w_;
  p.dout = d; p.hout = (h + 2 - 3) / sh + 1; p.wout = (w_ + 2 - 3) / sw + 1;
  p.cin = cin; p.cout = cout;
  p.lda = cin; p.ldw = (cout + 3) & ~3; p.ldo = cout;
  p.kd = p.kh = p.kw = 3;
  p.sd = 1; p.sh = sh; p.sw = sw;
  p.pd = p.ph = p.pw = 1;
  p.rv_rows = 1;
}

extern "C" int cs_conv3d_3x3x3_s111(const float* x, const float* w, const float* bias, float* out,
                                    int nb, int d, int h, int w_, int cin, int cout,
                                    cs_stream_t stream) {
  CsConvGemm p;
  fill_conv(p, x, w, bias, out, nb, d, h, w_, cin, cout, 1, 1);
  return cs_conv_gemm(&p, stream);
}

extern "C" int cs_conv3d_3x3x3_s122(const float* x, const float* w, const float* bias, float* out,
                                    int nb, int d, int h, int w_, int cin, int cout,
                                    cs_stream_t stream) {
  CsConvGemm p;
  fill_conv(p, x, w, bias, out, nb, d, h, w_, cin, cout, 2, 2);
  return cs_conv_gemm(&p, stream);
}

extern "C" int cs_gemm_tokens(const float* x, const float* w, const float* bias, const float* res,
                              float* out, int m, int k, int n, int act, cs_stream_t stream) {
  CsConvGemm p = CsConvGemm{};
  p.x = x; p.w = w; p.bias = bias; p.res = res; p.out = out;
  p.nb = m; p.din = p.hin = p.win = 1; p.dout = p.hout = p.wout = 1;
  p.cin = k; p.cout = n; p.lda = k; p.ldw = (n + 3) & ~3; p.ldo = n; p.ldr = n;
  p.kd = p.kh = p.kw = 1; p.sd = p.sh = p.sw = 1;
  p.act = act; p.rv_rows = 1;
  return cs_conv_gemm(&p, stream);
}

extern "C" int cs_relayout_weight(const float* w_torch, float* w_out, int cout, int cin, int taps,
                                  int cin_pad, int ldw, cs_stream_t stream) {
  if (!w_torch || !w_out || cout <= 0 || cin <= 0 || taps <= 0 || cin_pad < cin || ldw < cout)
    return CS_EINVAL;
  const int64_t total = (int64_t)taps * cin_pad * ldw;
  CS_LAUNCH(relayout_weight_kernel, dim3(cs_grid_for(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, w_torch, w_out, cout, cin, taps, cin_pad, ldw);
  CS_CHECK_LAUNCH();
  return CS_OK;
}

// r5: the Winograd-W weights of a 3x3x3 conv -- four packed images (position q-major), each in cs_pack_weight_f16x3's layout
// with the nine (kd, kh) taps: u_q over the kw taps [g0, (g0 + g1 + g2) / 2, (g0 - g1 + g2) / 2, g2], formed and split in fp64
namespace {
__global__ __launch_bounds__(256) void pack_f16x3_wino_kernel(const float* __restrict__ w, _Float16* __restrict__ wh,
                                                              _Float16* __restrict__ wl, int cout, int cin, int kg_per_tap,
                                                              float scale, int variant, int src_cin, int c0) {
  const int64_t per = 9LL * kg_per_tap * cout * 8;
  const int np = variant + 2;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < np * per; i += (int64_t)gridDim.x * blockDim.x) {
    const int q = (int)(i / per);
    int64_t t = i - q * per;
    const int j = (int)(t & 7);
    t >>= 3;
    const int n = (int)(t % cout);
    t /= cout;
    const int kg = (int)(t % kg_per_tap);
    const int tap = (int)(t / kg_per_tap);              // kd * 3 + kh
    const int c = kg * 8 + j;
    double u = 0.0;
    if (c < cin) {
      const float* g = w + ((int64_t)n * src_cin + c0 + c) * 27 + tap * 3;
      const double g0 = g[0], g1 = g[1], g2 = g[2];
      if (variant == 4)      // G of F(4,3): [1/4,0,0], [-1/6,-1/6,-1/6], [-1/6,1/6,-1/6], [1/24,1/12,1/6], [1/24,-1/12,1/6], [0,0,1]
        u = q == 0 ? g0 / 4.0 : q == 1 ? -(g0 + g1 + g2) / 6.0 : q == 2 ? (-g0 + g1 - g2) / 6.0
            : q == 3 ? g0 / 24.0 + g1 / 12.0 + g2 / 6.0 : q == 4 ? g0 / 24.0 - g1 / 12.0 + g2 / 6.0 : g2;
      else
        u = q == 0 ? g0 : q == 1 ? 0.5 * (g0 + g1 + g2) : q == 2 ? 0.5 * (g0 - g1 + g2) : g2;
    }
    const double v = u * (double)scale;
    const _Float16 h = (_Float16)v;
    wh[i] = h;
    wl[i] = (_Float16)(v - (double)h);
  }
}
}  // namespace

extern "C" int cs_pack_weight_f16x3_wino_v(const float* w_torch, void* w_hi, void* w_lo, int cout, int cin, float scale,
                                           int variant, int src_cin, int c0, cs_stream_t stream) {
  if (!w_torch || !w_hi || !w_lo || cout <= 0 || cin <= 0 || !(scale > 0.f) || (variant != 2 && variant != 4) || src_cin < 0 ||
      c0 < 0 || (src_cin && c0 + cin > src_cin))
    return CS_EINVAL;
  if (((uintptr_t)w_hi & 15) || ((uintptr_t)w_lo & 15)) return CS_EINVAL;
  const int kg_per_tap = ((cin + 15) / 16) * 2;
  const int64_t total = (int64_t)(variant + 2) * 9 * kg_per_tap * cout * 8;
  CS_LAUNCH(pack_f16x3_wino_kernel, dim3(cs_grid_for(total, 256, 256 * 32)), dim3(256), 0, (hipStream_t)stream, w_torch,
            (_Float16*)w_hi, (_Float16*)w_lo, cout, cin, kg_per_tap, scale, variant, src_cin ? src_cin : cin, c0);
  CS_CHECK_LAUNCH();
  return CS_OK;
}

extern "C" int cs_pack_weight_f16x3_wino(const float* w_torch, void* w_hi, void* w_lo, int cout, int cin, float scale,
                                         cs_stream_t stream) {
  return cs_pack_weight_f16x3_wino_v(w_torch, w_hi, w_lo, cout, cin, scale, 2, 0, 0, stream);
}

extern "C" int cs_abi_version(void) { return 18; }
