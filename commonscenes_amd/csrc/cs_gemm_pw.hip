// Persistent "ping-pong" pointwise GEMM for CS_MATH_F16X3 (tile code 5): the token / 1x1x1 GEMMs of the transformer
// blocks (K = 448 ... 2688, N a multiple of 224) whose K loop is too short to amortise a tile's epilogue.
//
// Why: with one 256x224 tile per workgroup the 28-chunk K loop of a 448-channel GEMM takes ~20 us and the tile's
// epilogue (229 KB written + 229 KB of residual read per CU, every CU at the same moment) another ~17 us with the
// matrix pipe idle (tools/gemm_ksweep.py): 155-290 TF/s where the 27-tap convs reach 400.  These GEMMs
// sit at the chip's balance point (75 flop per HBM byte), so the fix is overlap, not fewer bytes.
//
// How: one 512-thread workgroup per CU, persistent over tiles, split into two 4-wave GROUPS (one wave per SIMD each)
// that each own a 128x224 output tile and run half a period out of phase on a common barrier cadence:
//
//     slot:      | 0 1 2 ...                nk-1 | 0 1 2 ...                nk-1 |
//     group 0    |  K loop of tile a (MFMA)      |  epilogue of tile a (stores)  |  K loop of tile c ...
//     group 1    |  epilogue of tile z           |  K loop of tile b (MFMA)      |  epilogue of tile b ...
//
// Every slot is one workgroup-wide s_barrier.  The computing group does one 16-wide K chunk per slot exactly like
// conv_gemm_f16x3_kernel<1,7,4,1> (3-stage LDS-DMA ring, counted vmcnt, A split software-pipelined); the other group
// spreads its epilogue over the same slots -- residual / row-vector loads issued `stride` slots before they are
// consumed, so no slot ever waits on HBM latency -- and at the end prefetches the first two chunks of its next tile.
// The matrix pipe therefore always has exactly one wave per SIMD feeding it, and the output / residual traffic is
// spread evenly under the MFMA stream instead of arriving as a chip-wide burst.
//
// Results are bit-identical to the one-tile-per-workgroup kernels: same chunk order, same three-MFMA sequence per
// chunk, same epilogue expression.
#include "cs_f16x3.h"
#include <type_traits>

namespace {

using namespace cs16;

constexpr int BM = 128;                         // rows of a group's tile
constexpr int BN = 224;                         // columns (7 MFMA blocks of 32)
constexpr int WNB = 7;
constexpr int A_BYTES = BM * 64;                // raw fp32 [BM][16]
constexpr int B_BYTES = 2 * BN * 16;            // one fp16 image [2 k-groups][BN][8]
constexpr int STAGE = A_BYTES + 2 * B_BYTES;    // 22528
constexpr int NSTAGE = 3;
constexpr int RING = NSTAGE * STAGE;            // 67584 per group
constexpr int DUMP = 2 * RING;                  // surplus DMA wave-instructions land here
constexpr int VECS = DUMP + 1024;               // per group: bias[224] then rowvec[224] of the tile in its epilogue
constexpr int VEC_BYTES = 2 * 1024;
constexpr int LDS_BYTES = VECS + 2 * VEC_BYTES; // 140288 of the CU's 163840
constexpr int B_WI = BN / 32;                   // 7 wave-instructions per B image
constexpr int A_PW = 2;                         // A wave-instructions per wave per chunk (8 over 4 waves)
constexpr int B_PW = 4;                         // hi + lo: 14 over 4 waves, 2 surplus
constexpr int D = A_PW + B_PW;                  // DMA instructions per wave per chunk
constexpr int EP_ROWS = 4;                      // rows a wave stages per epilogue pass (2 accumulator registers x 2 halves)
constexpr int EP_BYTES = EP_ROWS * BN * 4;      // 3584 per wave, in the idle third ring stage
constexpr int NPASS = 8;
constexpr int MIN_NK = 28;                      // 8 passes x 3 slots + 4 slots of prefetch / hand-over

template <bool GEGLU>
__global__ __launch_bounds__(512, 2) void pw_gemm_f16x3_kernel(const CsConvGemm p, int M, int tiles_n, int T, int nk,
                                                              long long x_bytes, unsigned w_bytes, int iters,
                                                              int rv_shift, long long res_bytes) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int group = wave >> 2;                  // 0 / 1: the two tile owners
  const int wig = wave & 3;                     // wave in group: rows 32*wig .. 32*wig+31 of the tile
  const int l31 = lane & 31;
  const int half = lane >> 5;
  const int wm0 = wig * 32;
  unsigned char* const ring = smem + group * RING;

  // block -> slot in the tile sequence: block b runs on XCD b % 8; give every XCD a contiguous run of tiles so the
  // n-tiles that share a 128-row A slab (and neighbouring slabs) meet in one L2
  const int G = gridDim.x;
  const int wq = (G & 7) == 0 ? (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;

  const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t lrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w_lo, 0, w_bytes, 0x00020000);

  // ---- per-lane DMA constants (tile independent) ----
  // A wave-instruction v (0..7) of the group covers rows 16v .. 16v+15: row = 16v + lane/4, LDS slot q = lane&3
  // holds global 16-byte piece q ^ ((row>>2)&3)
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

  // the tile this group computes next / is computing (k*) and the one whose accumulators it holds (e*)
  int m0k = 0, n0k = 0, m0e = 0, n0e = 0;
  bool validk = false, valide = false;
  __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, 0, 0x00020000);
  auto setup_tile = [&](int t) {
    validk = t < T;
    const int tt = validk ? t : 0;
    const int tn = tt % tiles_n;
    m0k = (tt / tiles_n) * BM;
    n0k = tn * BN;
    // descriptor window based at the tile's first row: 32-bit offsets inside 128 rows, tensors of any size
    const long long skip = (long long)m0k * p.lda * 4;
    const long long left = x_bytes - skip;
    const unsigned win = left > 0xFFE00000LL ? 0xFFE00000u : (left > 0 ? (unsigned)left : 0u);
    xrs = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.x + skip), 0, win, 0x00020000);
  };
  auto issue_dma = [&](int cc, int stage) {
    unsigned char* st = ring + stage * STAGE;
    const bool live = cc < nk;
#pragma unroll
    for (int i = 0; i < A_PW; ++i) {
      const int c = cc * BKH + (int)a_piece[i];
      const unsigned off = (live && c < p.cin) ? a_rowoff[i] + (unsigned)c * 4u : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, st + (wig * A_PW + i) * 1024, 16, off, 0, 0, 0);
    }
    const unsigned kbase = (unsigned)(cc * 2 * p.cout + n0k) * 16u;
#pragma unroll
    for (int i = 0; i < B_PW; ++i) {
      const int v = wig * B_PW + i;                // wave-uniform
      const int img = v / B_WI;
      const unsigned off = (b_rel[i] == OOB || !live) ? OOB : b_rel[i] + kbase;
      unsigned char* dst = v < 2 * B_WI ? st + A_BYTES + img * B_BYTES + (v - img * B_WI) * 1024 : smem + DUMP;
      if (img == 1)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(lrs, dst, 16, off, 0, 0, 0);
      else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(hrs, dst, 16, off, 0, 0, 0);
    }
  };

  // fragment addressing (same maps as conv_gemm_f16x3_kernel)
  int a_frag[2];
  {
    const int row = wm0 + l31;
    const int s = (row >> 2) & 3;
    a_frag[0] = row * 64 + (((2 * half) ^ s) * 16);
    a_frag[1] = row * 64 + (((2 * half + 1) ^ s) * 16);
  }
  const int b_frag = A_BYTES + (half * BN + l31) * 16;
  auto load_a = [&](int st, h8& hi, h8& lo) {
    const unsigned char* s = ring + st * STAGE;
    const f32x4 x0 = *reinterpret_cast<const f32x4*>(s + a_frag[0]);
    const f32x4 x1 = *reinterpret_cast<const f32x4*>(s + a_frag[1]);
    split8(x0, x1, a_scale, hi, lo, amax);
  };

  f32x16 acc[WNB];
  h8 ah, al;
  int dcc = 0;
  auto step = [&](auto stage_c) {
    constexpr int stage = decltype(stage_c)::value;
    constexpr int nstage = (stage + 1) % NSTAGE;
    constexpr int dstage = (stage + 2) % NSTAGE;
    wait_vmcnt<B_PW>();                            // everything up to A(k+1) has landed for this wave
    __builtin_amdgcn_s_barrier();                  // ... and for every wave; the other group's slot boundary too
    const unsigned char* s = ring + stage * STAGE;
    h8 ah2, al2;
    // this wave is alone on its SIMD's matrix pipe (its partner is in the epilogue): fetch column block j+1's B
    // fragments before block j's MFMAs, so each LDS read has three MFMAs (96 cycles) of cover instead of one
    h8 bh = *reinterpret_cast<const h8*>(s + b_frag);
    h8 bl = *reinterpret_cast<const h8*>(s + b_frag + B_BYTES);
#pragma unroll
    for (int j = 0; j < WNB; ++j) {
      h8 bhn = bh, bln = bl;
      if (j + 1 < WNB) {
        bhn = *reinterpret_cast<const h8*>(s + b_frag + (j + 1) * 512);
        bln = *reinterpret_cast<const h8*>(s + b_frag + B_BYTES + (j + 1) * 512);
      }
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[j], 0, 0, 0);
      if (j == 0) {
        issue_dma(dcc, dstage);
        ++dcc;
        load_a(nstage, ah2, al2);
      }
      bh = bhn;
      bl = bln;
    }
    ah = ah2;
    al = al2;
  };

  // ---- epilogue pieces (the group that is NOT computing) ----
  constexpr int UPR = GEGLU ? BN / 8 : BN / 4;                   // float4 units per staged row: 28 / 56
  constexpr int UNITS = EP_ROWS * UPR;                           // 112 / 224
  constexpr int UPL = (UNITS + 63) / 64;                         // units per lane: 2 / 4
  // The tile's bias row and (to_out GEMMs) its row-vector row -- a 128-row tile lies inside one sample, the host
  // checks rv_rows % 128 == 0 -- are DMA'd into LDS at the first epilogue slot: no registers held, no load latency
  // in a later slot.  Residual rows are DMA'd `stride` slots ahead into ring stage 0 (idle until the prefetch of the
  // next tile at slot nk-3): one 4 KB slab per wave, unit u (16 bytes) of a pass at slab + 16 u.
  float* const ep = reinterpret_cast<float*>(ring + 2 * STAGE + wig * EP_BYTES);
  float* const ep_res = reinterpret_cast<float*>(ring + wig * 4096);
  __amdgpu_buffer_rsrc_t rrs_res = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, 0, 0x00020000);
  auto setup_res = [&]() {
    if (!GEGLU && p.res) {
      const long long skip = (long long)m0e * p.ldr * 4;
      const long long left = res_bytes - skip;
      const unsigned win = left > 0xFFE00000LL ? 0xFFE00000u : (left > 0 ? (unsigned)left : 0u);
      rrs_res = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.res + skip), 0, win, 0x00020000);
    }
  };
  float* const vec_bias = reinterpret_cast<float*>(smem + VECS + group * VEC_BYTES);
  float* const vec_rv = vec_bias + 256;
  float* const outp = p.out;
  const __amdgpu_buffer_rsrc_t brs =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.bias ? p.bias : p.out), 0, p.bias ? (unsigned)p.cout * 4u : 0u, 0x00020000);
  auto unit_row = [&](int q, int lrow) { return wm0 + 2 * (q & 1) + 8 * (q >> 1) + (lrow & 1) + 4 * (lrow >> 1); };
  auto epi_vectors = [&]() {
    // wave `wig` fetches floats 64*wig .. 64*wig+63 of the tile's 224 columns (4 bytes per lane); columns past cout
    // (the last wave's upper half) are outside the descriptor and read as zero
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
    }
  };
  auto epi_issue = [&](int q) {
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
  auto epi_finish_rt = [&](int q) {
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

  // ---- schedule ----
  const int stride = max(3, (nk - 4) / NPASS);     // slots between an epilogue pass's loads and its stores
  const int P = 2 * iters + 1;                     // phases; group g computes in the phases with (phase & 1) == g
  if (group == 0) {
    setup_tile(wq);
    if (validk) {
      issue_dma(0, 0);
      issue_dma(1, 1);
    }
    wait_vmcnt<D>();
  }
  __builtin_amdgcn_s_barrier();
  if (group == 0 && validk) load_a(0, ah, al);

  for (int ph = 0; ph < P; ++ph) {
    if ((ph & 1) == group) {
      // ------------------------------------------------ K phase ------------------------------------------------
      if (validk) {
#pragma unroll
        for (int j = 0; j < WNB; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        dcc = 2;
        for (int kc = 0; kc < nk; kc += NSTAGE) {
          step(std::integral_constant<int, 0>{});
          if (kc + 1 < nk) step(std::integral_constant<int, 1>{});
          if (kc + 2 < nk) step(std::integral_constant<int, 2>{});
        }
      } else {
        for (int s = 0; s < nk; ++s) __builtin_amdgcn_s_barrier();
      }
      m0e = m0k;
      n0e = n0k;
      valide = validk;
    } else {
      // ------------------------------------------------ E phase ------------------------------------------------
      // drain the two zero-fill prefetches the K loop issued past its end (they target ring stages the staging
      // below shares with the other waves of this group): wait here, the slot-0 barrier publishes it
      if (valide) wait_vmcnt<0>();
      const bool more = ph + 1 < P;
      int next_issue = 0, next_finish = stride, qi = 0, qf = 0;
      for (int s = 0; s < nk; ++s) {
        if (s == nk - 1) wait_vmcnt<D>();          // chunk 0 of the next tile has landed for this wave
        __builtin_amdgcn_s_barrier();
        if (valide) {
          if (s == 0) {
            epi_vectors();                         // lands long before the first finish, `stride` (>= 3) slots on
            setup_res();
          }
          if (s == next_finish && qf < NPASS) {
            epi_finish_rt(qf);
            ++qf;
            next_finish += stride;
          }
          if (s == next_issue && qi < NPASS) {
            epi_issue(qi);
            ++qi;
            next_issue += stride;
          }
        }
        if (s == nk - 3) {
          setup_tile(more ? (2 * ((ph + 1) >> 1) + group) * G + wq : T);
          if (validk) issue_dma(0, 0);
        } else if (s == nk - 2) {
          if (validk) issue_dma(1, 1);
        } else if (s == nk - 1) {
          if (validk) load_a(0, ah, al);
        }
      }
      valide = false;
    }
  }
  wait_vmcnt<0>();
  if (p.status && amax >= 65504.f) atomicOr(p.status, CS_STATUS_F16X3_OVERFLOW);
}

}  // namespace

// Does cs_conv_gemm's auto-selection (tile 0) take the ping-pong kernel?  Only where it measured faster than the
// one-tile-per-workgroup kernels on the MI355X (tools/gemm_1tap.py, profiles/r02_pingpong_1tap.txt): short K loops
// (K <= 672) with a plain epilogue and enough tiles for two full rounds -- the C x C token GEMMs and the level-0
// skip projections.  Long-K shapes are latency-bound on the activation stream (two chunks in flight per group are
// too few) and stay on the 256x224 tile.  Mirrors commonscenes_amd/ops.py::tile_for.
bool cs_pw_gemm_f16x3_preferred(const CsConvGemm& p, int64_t M) {
  const int nk = (p.cin + 15) / 16;
  const int64_t T = ((M + BM - 1) / BM) * (p.cout / BN);
  return nk <= 42 && T >= 1024 && p.cout <= 448 && p.act != CS_ACT_GEGLU;
}

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
  const int iters = (int)((T + 2 * G - 1) / (2 * G));
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
              (unsigned)w_bytes, iters, rv_shift, (long long)res_bytes);
  else
    CS_LAUNCH(pw_gemm_f16x3_kernel<false>, dim3(G), dim3(512), 0, s, p, M, tiles_n, (int)T, nk, (long long)x_bytes,
              (unsigned)w_bytes, iters, rv_shift, (long long)res_bytes);
  CS_CHECK_LAUNCH();
  return CS_OK;
}
