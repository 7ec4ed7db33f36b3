// Persistent "ping-pong" pointwise GEMM for CS_MATH_F16X3 (tile code 5): the token / 1x1x1 GEMMs of the transformer
// blocks (K = 448 ... 2688, N a multiple of 224).
//
// Why.  With one 256x224 tile per workgroup these GEMMs ran at 155-290 TF/s where the 27-tap convs reach 400
// (profiles/r02_c_gemm_table_before_pingpong.txt), for two reasons a conv does not have:
//   * short K loops: a 448-channel GEMM's 28 chunks take ~20 us, and the tile's prologue (first operands from HBM) and
//     epilogue (229 KB written + 229 KB of residual read per CU, every CU at the same moment) another ~20 us with the
//     matrix pipe idle;
//   * long K loops are latency-bound on the activation stream: a pointwise GEMM reads every A byte from HBM exactly
//     once (a conv re-touches its rows 27 times from L1/L2), and two 16 KB chunks in flight per CU cover ~1 us of a
//     ~2 us loaded HBM latency.
//
// How.  One 512-thread workgroup per CU, persistent over its tiles, is two 4-wave GROUPS (one wave per SIMD each) that
// own alternate 128x224 tiles and run half a period out of phase on a common barrier cadence (one s_barrier per
// 16-wide K chunk = "slot"):
//
//     slot:      | 0 1 2 ...                nk-1 | 0 1 2 ...                nk-1 |
//     group 0    |  K loop of tile 0 (MFMA)      |  epilogue of tile 0 (stores)  |  K loop of tile 2 ...
//     group 1    |  (idle)                       |  K loop of tile 1 (MFMA)      |  epilogue of tile 1 ...
//
//   * The operands of ALL tiles form one continuous chunk stream through a shared LDS ring: six A stages (8 KB each:
//     five chunks = 40 KB per CU in flight, enough for HBM latency at full bandwidth) and three B stages (14 KB,
//     L2-resident weights).  The computing group issues the stream's LDS-DMAs -- chunk c+5's A and chunk c+2's B while it
//     multiplies chunk c -- straight across tile boundaries, so the next tile (the OTHER group's) finds its first
//     chunks in LDS: no prologue.
//   * The group that is not computing spreads its tile's epilogue over the same slots: accumulators -> per-wave LDS
//     staging -> float4 rows; bias / row-vector rows and the residual rows arrive by LDS-DMA `stride` slots before they
//     are consumed, so no slot ever waits on HBM latency and no registers are held across slots.
//   The matrix pipe therefore always has exactly one wave per SIMD feeding it, and the output / residual traffic is
//   spread evenly under the MFMA stream instead of arriving as a chip-wide burst.
//
// Results are bit-identical to the one-tile-per-workgroup kernels: same chunk order, same three-MFMA sequence per
// chunk, same epilogue expression.
#include "cs_f16x3.h"
#include <type_traits>

namespace {

using namespace cs16;

#ifndef PW_ABLATE
#define PW_ABLATE 0   // debug builds, timing only (results are wrong): 1 = no epilogue work, 2 = no stream DMAs,
                      // 4 = no sched_barrier pinning, 8 = no A split (load_a skipped), 16 = no MFMAs
#endif

constexpr int BM = 128;                         // rows of a group's tile
constexpr int BN = 224;                         // columns (7 MFMA blocks of 32)
constexpr int WNB = 7;
constexpr int A_BYTES = BM * 64;                // raw fp32 [BM][16]                       8192
constexpr int B_BYTES = 2 * BN * 16;            // one fp16 image [2 k-groups][BN][8]      7168
constexpr int NA = 6, NB = 3;                   // ring depths
constexpr int DA = NA - 1, DB = NB - 1;         // prefetch distances (chunks)
constexpr int A_RING = 0;
constexpr int B_RING = A_RING + NA * A_BYTES;                 // 49152
constexpr int EP_STAGE = B_RING + NB * 2 * B_BYTES;           // 92160: per wave 4 rows x 224 floats (3584 B) x 8 waves
constexpr int EP_BYTES = 4 * BN * 4;
constexpr int RES_STAGE = EP_STAGE + 8 * EP_BYTES;            // 120832: per wave 4 KB of residual rows x 8 waves
constexpr int VECS = RES_STAGE + 8 * 4096;                    // 153600: per group bias[256] | rowvec[256]
constexpr int DUMP = VECS + 2 * 2048;                         // 157696: surplus DMA wave-instructions land here
constexpr int LDS_BYTES = DUMP + 1024;                        // 158720 of the CU's 163840
constexpr int B_WI = BN / 32;                   // 7 wave-instructions per B image
constexpr int A_PW = 2;                         // A wave-instructions per wave per chunk (8 over 4 waves)
constexpr int B_PW = 4;                         // hi + lo: 14 over 4 waves, 2 surplus
constexpr int D = A_PW + B_PW;                  // DMA instructions per wave per slot
constexpr int NPASS = 8;                        // epilogue passes per tile (2 accumulator registers x 7 blocks each)
constexpr int MIN_NK = 28;                      // 2 hand-over slots + 8 passes x 3 slots + 2

template <bool GEGLU>
__global__ __launch_bounds__(512, 2) void pw_gemm_f16x3_kernel(const CsConvGemm p, int M, int tiles_n, int T, int nk,
                                                              long long x_bytes, unsigned w_bytes, int rv_shift,
                                                              long long res_bytes) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int group = wave >> 2;                  // 0 / 1: owners of the even / odd tiles of this workgroup
  const int wig = wave & 3;                     // wave in group: rows 32*wig .. 32*wig+31 of the tile
  const int l31 = lane & 31;
  const int half = lane >> 5;
  const int wm0 = wig * 32;

  // block -> slot in the tile sequence: block b runs on XCD b % 8; give every XCD a contiguous run of tiles so the
  // n-tiles that share a 128-row A slab (and neighbouring slabs) meet in one L2
  const int G = gridDim.x;
  const int wq = (G & 7) == 0 ? (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int J = wq < T ? (T - wq + G - 1) / G : 0;       // tiles of this workgroup: wq, wq + G, ...
  if (J == 0) return;

  const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t lrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w_lo, 0, w_bytes, 0x00020000);

  // ---- per-lane DMA constants (tile independent) ----
  // A wave-instruction v (0..7) covers rows 16v .. 16v+15: row = 16v + lane/4, LDS slot q = lane&3 holds global
  // 16-byte piece q ^ ((row>>2)&3)
  unsigned a_rowoff[A_PW];
  unsigned a_piece[A_PW];
#pragma unroll
  for (int i = 0; i < A_PW; ++i) {
    const int row = 16 * (wig * A_PW + i) + (lane >> 2);
    a_rowoff[i] = (unsigned)row * ((unsigned)p.lda * 4u);
    a_piece[i] = (unsigned)(((lane & 3) ^ ((row >> 2) & 3)) * 4);
  }
  unsigned b_rel[B_PW];                         // byte offset inside a chunk's [2][cout][8] slab (without n0), or OOB
#pragma unroll
  for (int i = 0; i < B_PW; ++i) {
    const int v = wig * B_PW + i;
    const int img = v / B_WI;
    const int u = (v - img * B_WI) * 64 + lane;
    const int kg = u / BN;
    const int n = u - kg * BN;
    b_rel[i] = v < 2 * B_WI ? (unsigned)((kg * p.cout + n) * 16) : OOB;
  }
  const float a_scale = p.a_scale;
  float amax = 0.f;

  // ---- the chunk stream: two cursors (next A chunk / next B chunk to fetch), kept in step by BOTH groups ----
  auto tile_m0 = [&](int j) { return ((j * G + wq) / tiles_n) * BM; };
  auto tile_n0 = [&](int j) { return ((j * G + wq) % tiles_n) * BN; };
  int aj = 0, akc = 0, bj = 0, bkc = 0;         // tile (index into this workgroup's sequence) and chunk of each cursor
  int sa_w = 0, sb_w = 0;                       // ring stages the cursors write next
  __amdgpu_buffer_rsrc_t xrs;
  int bn0 = 0;
  auto set_a_tile = [&]() {
    // descriptor window based at the tile's first row: 32-bit offsets inside 128 rows, tensors of any size
    const long long skip = aj < J ? (long long)tile_m0(aj) * p.lda * 4 : x_bytes;
    const long long left = x_bytes - skip;
    const unsigned win = left > 0xFFE00000LL ? 0xFFE00000u : (left > 0 ? (unsigned)left : 0u);
    xrs = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.x + (left > 0 ? skip : 0)), 0, win, 0x00020000);
  };
  auto set_b_tile = [&]() { bn0 = bj < J ? tile_n0(bj) : 0; };
  set_a_tile();
  set_b_tile();
  auto issue_a = [&](bool doit) {               // fetch the A cursor's chunk into stage sa_w, then advance
    if (doit) {
      unsigned char* st = smem + A_RING + sa_w * A_BYTES;
#pragma unroll
      for (int i = 0; i < A_PW; ++i) {
        const int c = akc * BKH + (int)a_piece[i];
        const unsigned off = (aj < J && c < p.cin) ? a_rowoff[i] + (unsigned)c * 4u : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, st + (wig * A_PW + i) * 1024, 16, off, 0, 0, 0);
      }
    }
    sa_w = sa_w + 1 == NA ? 0 : sa_w + 1;
    if (++akc == nk) {
      akc = 0;
      ++aj;
      set_a_tile();
    }
  };
  auto issue_b = [&](bool doit) {
    if (doit) {
      unsigned char* st = smem + B_RING + sb_w * 2 * B_BYTES;
      const unsigned kbase = (unsigned)(bkc * 2 * p.cout + bn0) * 16u;
#pragma unroll
      for (int i = 0; i < B_PW; ++i) {
        const int v = wig * B_PW + i;                // wave-uniform
        const int img = v / B_WI;
        const unsigned off = (b_rel[i] == OOB || bj >= J) ? OOB : b_rel[i] + kbase;
        unsigned char* dst = v < 2 * B_WI ? st + img * B_BYTES + (v - img * B_WI) * 1024 : smem + DUMP;
        if (img == 1)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(lrs, dst, 16, off, 0, 0, 0);
        else
          __builtin_amdgcn_raw_ptr_buffer_load_lds(hrs, dst, 16, off, 0, 0, 0);
      }
    }
    sb_w = sb_w + 1 == NB ? 0 : sb_w + 1;
    if (++bkc == nk) {
      bkc = 0;
      ++bj;
      set_b_tile();
    }
  };

  // fragment addressing (same maps as conv_gemm_f16x3_kernel); ring stage added per slot
  int a_frag[2];
  {
    const int row = wm0 + l31;
    const int s = (row >> 2) & 3;
    a_frag[0] = A_RING + row * 64 + (((2 * half) ^ s) * 16);
    a_frag[1] = A_RING + row * 64 + (((2 * half + 1) ^ s) * 16);
  }
  const int b_frag = B_RING + (half * BN + l31) * 16;
  int sa_r = 0, sb_r = 0;                       // ring stages of the chunk consumed in the current slot
  auto load_a = [&](int stage, h8& hi, h8& lo) {
    const f32x4 x0 = *reinterpret_cast<const f32x4*>(smem + a_frag[0] + stage * A_BYTES);
    const f32x4 x1 = *reinterpret_cast<const f32x4*>(smem + a_frag[1] + stage * A_BYTES);
    split8(x0, x1, a_scale, hi, lo, amax);
  };

  f32x16 acc[WNB];
  h8 ah, al;

  // ---- epilogue pieces (the group that is NOT computing) ----
  constexpr int UPR = GEGLU ? BN / 8 : BN / 4;                   // float4 units per staged row: 28 / 56
  constexpr int UNITS = 4 * UPR;                                 // 112 / 224
  constexpr int UPL = (UNITS + 63) / 64;                         // units per lane: 2 / 4
  float* const ep = reinterpret_cast<float*>(smem + EP_STAGE + wave * EP_BYTES);
  float* const ep_res = reinterpret_cast<float*>(smem + RES_STAGE + wave * 4096);
  float* const vec_bias = reinterpret_cast<float*>(smem + VECS + group * 2048);
  float* const vec_rv = vec_bias + 256;
  float* const outp = p.out;
  const __amdgpu_buffer_rsrc_t brs =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.bias ? p.bias : p.out), 0, p.bias ? (unsigned)p.cout * 4u : 0u, 0x00020000);
  __amdgpu_buffer_rsrc_t rrs_res = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, 0, 0x00020000);
  int m0e = 0, n0e = 0;
  auto unit_row = [&](int q, int lrow) { return wm0 + 2 * (q & 1) + 8 * (q >> 1) + (lrow & 1) + 4 * (lrow >> 1); };
  auto epi_vectors = [&]() {
    // The tile's bias row and (to_out GEMMs) its row-vector row -- a 128-row tile lies inside one sample, the host checks
    // rv_rows % 128 == 0 -- by LDS-DMA: wave `wig` fetches floats 64 wig .. 64 wig + 63 of the 224 columns
    const unsigned col = (unsigned)(64 * wig + lane);
    const unsigned off = col < (unsigned)BN ? (unsigned)(n0e + (int)col) * 4u : OOB;
    if (p.bias) __builtin_amdgcn_raw_ptr_buffer_load_lds(brs, vec_bias + 64 * wig, 4, off, 0, 0, 0);
    if constexpr (!GEGLU) {
      if (p.rowvec) {
        const int rvr = rv_shift >= 0 ? (m0e >> rv_shift) : (m0e / p.rv_rows);
        const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(p.rowvec + (int64_t)rvr * p.ldrv), 0, (unsigned)p.cout * 4u, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rrs, vec_rv + 64 * wig, 4, off, 0, 0, 0);
      }
      if (p.res) {
        const long long skip = (long long)m0e * p.ldr * 4;
        const long long left = res_bytes - skip;
        const unsigned win = left > 0xFFE00000LL ? 0xFFE00000u : (left > 0 ? (unsigned)left : 0u);
        rrs_res = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.res + skip), 0, win, 0x00020000);
      }
    }
  };
  auto epi_issue = [&](int q) {                  // pass q's residual rows -> this wave's LDS slab (unit u at 16 u)
    if constexpr (!GEGLU) {
      if (p.res) {
#pragma unroll
        for (int k = 0; k < UPL; ++k) {
          const int u = lane + 64 * k;               // units past 224 (and rows past M) fetch nothing: zero fill
          const int lrow = u / UPR;
          const int c4 = u - lrow * UPR;
          const int row = unit_row(q, lrow & 3);
          const unsigned off = (u < UNITS && m0e + row < M) ? (unsigned)(row * p.ldr + n0e + 4 * c4) * 4u : OOB;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rrs_res, ep_res + 256 * k, 16, off, 0, 0, 0);
        }
      }
    }
  };
  // accumulator registers 2q, 2q+1 of every column block -> the wave's staging rows (the only part of a pass that needs
  // a compile-time q: register indices); everything after it addresses by the run-time q, so there is ONE copy of it
  auto epi_stage = [&](auto q_c) {
    constexpr int q = decltype(q_c)::value;
#pragma unroll
    for (int j = 0; j < WNB; ++j)
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) ep[(rr + 2 * half) * BN + 32 * j + l31] = acc[j][2 * q + rr] * p.acc_scale;
  };
  auto epi_finish = [&](int q) {
    if (!GEGLU && p.res) wait_vmcnt<0>();          // this pass's residual rows (issued `stride` slots ago) are in LDS
    switch (q) {
      case 0: epi_stage(std::integral_constant<int, 0>{}); break;
      case 1: epi_stage(std::integral_constant<int, 1>{}); break;
      case 2: epi_stage(std::integral_constant<int, 2>{}); break;
      case 3: epi_stage(std::integral_constant<int, 3>{}); break;
      case 4: epi_stage(std::integral_constant<int, 4>{}); break;
      case 5: epi_stage(std::integral_constant<int, 5>{}); break;
      case 6: epi_stage(std::integral_constant<int, 6>{}); break;
      default: epi_stage(std::integral_constant<int, 7>{}); break;
    }
    // same-wave LDS ops are ordered; the compiler waits on lgkmcnt before the reads below
#pragma unroll
    for (int k = 0; k < UPL; ++k) {
      const int u = lane + 64 * k;
      if (u < UNITS) {
        const int lrow = u / UPR;
        const int c4 = u - lrow * UPR;
        const int m = m0e + unit_row(q, lrow);
        if (m < M) {
          if constexpr (GEGLU) {
            // columns of the tile = [x (112) | gate (112)] (ops.pack_geglu_weight): out = (x + b_x) * gelu(gate + b_g)
            f32x4 xv = *reinterpret_cast<const f32x4*>(ep + lrow * BN + 4 * c4);
            f32x4 gv = *reinterpret_cast<const f32x4*>(ep + lrow * BN + BN / 2 + 4 * c4);
            if (p.bias) {
              xv += *reinterpret_cast<const f32x4*>(vec_bias + 4 * c4);
              gv += *reinterpret_cast<const f32x4*>(vec_bias + BN / 2 + 4 * c4);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) xv[e] = xv[e] * cs_gelu(gv[e]);
            *reinterpret_cast<f32x4*>(outp + (int64_t)m * p.ldo + n0e / 2 + 4 * c4) = xv;
          } else {
            f32x4 v = *reinterpret_cast<const f32x4*>(ep + lrow * BN + 4 * c4);
            if (p.bias) v += *reinterpret_cast<const f32x4*>(vec_bias + 4 * c4);
            if (p.rowvec) v += *reinterpret_cast<const f32x4*>(vec_rv + 4 * c4);
            if (p.act != CS_ACT_NONE) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = cs_act(v[e], p.act);
            }
            if (p.res) v += *reinterpret_cast<const f32x4*>(ep_res + 4 * u);
            *reinterpret_cast<f32x4*>(outp + (int64_t)m * p.ldo + n0e + 4 * c4) = v;
          }
        }
      }
    }
  };

  // ---- prologue: the stream's head (A chunks 0..4, B chunks 0..1), fetched by group 0, the first to compute ----
#pragma unroll 1
  for (int i = 0; i < DA; ++i) issue_a(group == 0);
#pragma unroll 1
  for (int i = 0; i < DB; ++i) issue_b(group == 0);
  if (group == 0) wait_vmcnt<B_PW>();             // everything but B(1) has landed: A(0..4), B(0)
  __builtin_amdgcn_s_barrier();
  if (group == 0) load_a(0, ah, al);

  // ---- phases: in phase ph the owner of tile ph computes it, the other group finishes tile ph - 1 ----
  const int stride = max(3, (nk - 4) / NPASS);     // slots between an epilogue pass's loads and its stores
  for (int ph = 0; ph <= J; ++ph) {
    if ((ph & 1) == group && ph < J) {
      // ------------------------------------------------ K phase ------------------------------------------------
#pragma unroll
      for (int j = 0; j < WNB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll 1
      for (int s = 0; s < nk; ++s) {
        // Of this wave's stream fetches only the newest slot's may still fly: B(c) (fetched two slots ago) and
        // A(c+1) (four slots ago) have landed.  During the first two slots of a phase they are the OTHER group's
        // fetches: it waits for them in its own slots 0 and 1 (below).
        wait_vmcnt<D>();
        __builtin_amdgcn_s_barrier();
        const unsigned char* sb = smem + sb_r * 2 * B_BYTES;
        const int sa_n = sa_r + 1 == NA ? 0 : sa_r + 1;
        h8 ah2, al2;
        // this wave is alone on its SIMD's matrix pipe (its partner is in the epilogue): fetch column block j+1's B
        // fragments before block j's MFMAs, so each LDS read has three MFMAs (96 cycles) of cover
        h8 bh = *reinterpret_cast<const h8*>(sb + b_frag);
        h8 bl = *reinterpret_cast<const h8*>(sb + b_frag + B_BYTES);
#pragma unroll
        for (int j = 0; j < WNB; ++j) {
          h8 bhn = bh, bln = bl;
          if (j + 1 < WNB) {
            bhn = *reinterpret_cast<const h8*>(sb + b_frag + (j + 1) * 512);
            bln = *reinterpret_cast<const h8*>(sb + b_frag + B_BYTES + (j + 1) * 512);
          }
          if (!(PW_ABLATE & 4)) __builtin_amdgcn_sched_barrier(0);       // keep the prefetch above the MFMAs it covers
          if (!(PW_ABLATE & 16)) {
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[j], 0, 0, 0);
          } else { acc[j][0] += (float)bh[0] + (float)bl[0]; }
          if (j == 0) {
            issue_a(!(PW_ABLATE & 2));               // chunk c + 5 (the stage chunk c - 1's fragments left two slots ago)
            issue_b(!(PW_ABLATE & 2));               // chunk c + 2 (the stage chunk c - 1 left at this slot's barrier)
            if (!(PW_ABLATE & 8)) load_a(sa_n, ah2, al2); else { ah2 = ah; al2 = al; }   // chunk c + 1's fragments
          }
          bh = bhn;
          bl = bln;
        }
        ah = ah2;
        al = al2;
        sa_r = sa_n;
        sb_r = sb_r + 1 == NB ? 0 : sb_r + 1;
      }
      m0e = tile_m0(ph);
      n0e = tile_n0(ph);
    } else {
      // ------------------------------------------------ E phase ------------------------------------------------
      const bool valide = (ph & 1) != group && ph >= 1;     // this group computed tile ph - 1 in the last phase
      const bool next_mine = (ph & 1) != group && ph + 1 < J;   // ... and computes tile ph + 1 in the next one
      int next_issue = 2, next_finish = 2 + stride, qi = 0, qf = 0;
#pragma unroll 1
      for (int s = 0; s < nk; ++s) {
        // hand-over: the last stream fetches this group issued (as the computing group of the previous phase) feed the
        // other group's first slots: B(c0) at its slot 0; B(c0+1), A(c0+2..4) at its slot 1
        if (s == 0) wait_vmcnt<D>();
        if (s == 1) wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        issue_a(false);                              // keep the cursors and ring positions in step with the stream
        issue_b(false);
        sa_r = sa_r + 1 == NA ? 0 : sa_r + 1;
        sb_r = sb_r + 1 == NB ? 0 : sb_r + 1;
        if (valide && !(PW_ABLATE & 1)) {
          if (s == next_finish && qf < NPASS) {
            epi_finish(qf);
            ++qf;
            next_finish += stride;
          }
          if (s == next_issue && qi < NPASS) {
            if (qi == 0) epi_vectors();              // bias / row-vector rows land long before the first finish
            epi_issue(qi);
            ++qi;
            next_issue += stride;
          }
        }
        if (s == nk - 1 && next_mine) load_a(sa_r, ah, al);   // first chunk of my next tile (landed: A(c+1) rule)
      }
    }
  }
  wait_vmcnt<0>();
  if (p.status && amax >= 65504.f) atomicOr(p.status, CS_STATUS_F16X3_OVERFLOW);
}

}  // namespace

// Does cs_conv_gemm's auto-selection (tile 0) take the ping-pong kernel?  No: measured on the MI355X
// (tools/gemm_1tap.py, tools/pw_ablate.sh; profiles/r02_pingpong_*.txt) it loses to the 256x224 tile once that tile's
// epilogue prefetches its residual rows (cs_gemm_f16x3.hip "pipelined residual epilogue").  What the ablations showed:
// the 128-row tiles double the weight-operand LDS-DMA traffic per flop (the chunk stream alone -- no MFMA, no
// epilogue -- takes 131-218 us on shapes the tile kernel finishes in 340-380 us), one wave per SIMD keeps the matrix
// pipe ~55 % busy even with no DMA at all, and the two do not overlap well enough to pay for the hidden epilogue.
// The real loss of the tile kernels was not the epilogue's traffic but its 28 serialized residual-load latencies.
// The kernel stays available as tile = 5 (bit-identical, tested) for further work on it.
bool cs_pw_gemm_f16x3_preferred(const CsConvGemm&, int64_t) { return false; }

// Can the ping-pong kernel run this (validated) descriptor at all (explicit tile = 5)?
bool cs_pw_gemm_f16x3_applicable(const CsConvGemm& p, int64_t M) {
  if (p.math != CS_MATH_F16X3 || p.a_format != 0 || p.splitk > 1) return false;
  if (p.kd != 1 || p.kh != 1 || p.kw != 1 || p.sd != 1 || p.sh != 1 || p.sw != 1 || p.ud || p.uh || p.uw) return false;
  if (p.din != p.dout || p.hin != p.hout || p.win != p.wout) return false;
  if (p.cout % BN || p.scale) return false;
  const int nk = (p.cin + 15) / 16;
  if (nk < MIN_NK) return false;
  const int64_t T = ((M + BM - 1) / BM) * (p.cout / BN);
  if (T < 384) return false;                       // too few tiles to keep both groups of 256 workgroups busy
  auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
  if ((p.ldo & 3) || !al16(p.out) || !al16(p.w_lo) || (p.bias && !al16(p.bias)) ||
      (p.rowvec && ((p.ldrv & 3) || !al16(p.rowvec))) || (p.res && ((p.ldr & 3) || !al16(p.res))))
    return false;
  if (p.act == CS_ACT_GEGLU && (p.rowvec || p.res)) return false;
  if (p.rowvec && (p.rv_rows % BM)) return false;  // a 128-row tile must lie inside one row-vector entry
  return true;
}

int cs_pw_gemm_f16x3_launch(const CsConvGemm& p_in, int M, hipStream_t s) {
  CsConvGemm p = p_in;
  if (p.a_scale == 0.f) p.a_scale = A_SCALE_DEFAULT;
  if (!cs_pw_gemm_f16x3_applicable(p, M) || !p.w_lo || !(p.acc_scale > 0.f) || !(p.a_scale > 0.f)) return CS_EINVAL;
  const int tiles_m = (M + BM - 1) / BM;
  const int tiles_n = p.cout / BN;
  const int64_t T = (int64_t)tiles_m * tiles_n;
  if (T > 0x3fffffffLL) return CS_EINVAL;
  int G = 256;
  if (T < 2 * G) G = (int)((T + 1) / 2);           // every group gets a tile
  const int nk = (p.cin + 15) / 16;
  const int kg_per_tap = nk * 2;
  const int64_t x_bytes = ((int64_t)(M - 1) * p.lda + p.cin) * 4;
  const int64_t w_bytes = (int64_t)kg_per_tap * p.cout * 16;
  if (w_bytes > 0xFFE00000LL || (int64_t)BM * p.lda * 4 > 0x7FF00000LL) return CS_EINVAL;
  int rv_shift = -1;
  if (p.rowvec && p.rv_rows > 0 && (p.rv_rows & (p.rv_rows - 1)) == 0) {
    rv_shift = 0;
    while ((1 << rv_shift) < p.rv_rows) ++rv_shift;
  }
  const int64_t res_bytes = p.res ? ((int64_t)(M - 1) * p.ldr + p.cout) * 4 : 0;
  if (p.res && (int64_t)BM * p.ldr * 4 > 0x7FF00000LL) return CS_EINVAL;
  if (p.act == CS_ACT_GEGLU)
    CS_LAUNCH(pw_gemm_f16x3_kernel<true>, dim3(G), dim3(512), 0, s, p, M, tiles_n, (int)T, nk, (long long)x_bytes,
              (unsigned)w_bytes, rv_shift, (long long)res_bytes);
  else
    CS_LAUNCH(pw_gemm_f16x3_kernel<false>, dim3(G), dim3(512), 0, s, p, M, tiles_n, (int)T, nk, (long long)x_bytes,
              (unsigned)w_bytes, rv_shift, (long long)res_bytes);
  CS_CHECK_LAUNCH();
  return CS_OK;
}
