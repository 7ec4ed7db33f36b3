// Implicit-GEMM conv3d / linear with fp32 operands carried as fp16 hi/lo pairs on the fp16 MFMA
// (v_mfma_f32_32x32x16_f16) -- CS_MATH_F16X3.
//
// Why: gfx950's fp32-input MFMA runs at the vector rate (157 TF); its fp16 MFMA runs at 2.5 PF.  A fp32
// value v*2^s split as hi = fp16(v'), lo = fp16(v' - hi) keeps 22 mantissa bits, every fp16 x fp16 product is
// exact in the fp32 accumulator, and  a.w ~= a_hi.w_hi + a_hi.w_lo + a_lo.w_hi  drops only the 2^-22 term.
// Three fp16 MFMAs per K=16 step cost 96 SIMD cycles against 512 for the fp32-input MFMA: a 5.3x higher
// matrix-pipe ceiling at ~fp32 accuracy (measured error vs fp64 is reported by tests/test_f16x3_gpu.py).
//
// Data movement is ALL LDS-DMA (buffer_load ... lds): no VGPR staging, no ds_write, no data-dependent
// branch in the K loop.
//   * weights are split offline (cs_pack_weight_f16x3) into hi / lo images laid out [tap][k/8][n][8 halves]
//     = the MFMA B-fragment order, so a B tile is a straight 16-byte-per-lane DMA;
//   * activations arrive as raw fp32 rows ([m][16 ch] = 64 B, 16-byte pieces XOR-swizzled by (m>>2)&3 on the
//     SOURCE address so the fragment reads are conflict-free) and are split into hi / lo fp16
//     (scale a_scale, default 2^4)
//     by the consuming wave inside the previous chunk's MFMA stream (VALU work that overlaps the other wave's MFMAs);
//   * zero padding (out-of-image taps, rows past M, channels past cin, columns past cout) is an out-of-range
//     buffer offset: the buffer unit writes zeros, the loop stays one basic block.
// Pipeline: 3-stage LDS ring, one raw s_barrier per K-chunk, counted vmcnt (DMAs stay in flight across it):
//     wait vmcnt(B_PW) [A(k+1) and older landed] ; s_barrier ; MFMAs on chunk k || issue DMA(k+2), split A(k+1)
// K order: channel chunk OUTER, tap INNER (the 27 taps of one 16-channel chunk re-touch only this tile's rows
// + halo, so they hit L1/L2 instead of re-streaming the activation tensor per tap).
// fp16 32x32x16 operand map: lane l holds row/col l&31 and k = 8*(l>>5) .. 8*(l>>5)+7; C/D as for fp32.
#include <cstdlib>
#include <cstring>
#include "cs_f16x3.h"
#include <type_traits>
#include <utility>

namespace {

using cs16::h8;
using cs16::split8;
using cs16::wait_vmcnt;
using cs16::BKH;
using cs16::A_SCALE_DEFAULT;
using cs16::OOB;
constexpr int MAX_TAPS = 27;
#ifndef CS_ABLATE
#define CS_ABLATE 0   // debug builds, timing only (results are wrong): 1 = no DMA issue, 2 = DMAs fetch nothing (all
                      // offsets out of range -> zero fill), 4 = no vmcnt wait in the loop, 8 = no barrier,
                      // 16 = s_setprio around the MFMAs, 32 = scalar epilogue, 64 = all fetches from one 16 KB window,
                      // 1024 = no epilogue (K loop + prologue only), 2048 = no K loop (prologue + epilogue only),
                      // 4096 / 8192 = slab kernel reads every second / only the first B fragment pair from LDS
                      // (r6, the piped epilogue's own budget) 16384 = its global stores dropped, 32768 = the fused gate
                      // multiplies by the raw gate column instead of its GELU
#endif

// PRE = the activations arrive already split: p.x / p.x_lo are fp16 hi / lo images [rows][lda halves] written by
// the producer (cs_groupnorm_apply_split16) with the a_scale factor applied -- the split is then done once per
// element instead of once per tap per N-tile, and the K loop carries no conversion VALU at all.
//
// SLAB (3x3x3, stride 1, "same" padding, no upsampling, 256-row tiles, one K slice): consecutive output voxels read
// consecutive source rows, and the nine (kh, kw) taps of one kd read the SAME rows shifted by kh*W + kw.  The A operand
// is therefore staged once per (kd, 16-channel chunk) as a slab of BM + 2W + 2 source rows, and the nine taps' fragment
// reads address it at their row shift; out-of-volume taps are zeroed per lane from a 27-bit validity mask held in a
// register.  That replaces nine 16 KB gathers (16 scattered 64-byte pieces per wave-instruction, the expensive half of
// the DMA stream: dropping them in a timing-only build moved the conv shapes from 385-388 to 448-450 TF/s,
// tools/slab_whatif.sh) by one ~18 KB sequential one.  The chunk order, and with it every accumulation order, is the
// same as without the slab: results are bit-identical.
//
// PAIR (r3, a_format = 2): the activations arrive as the INTERLEAVED operand pair -- per row and 16-channel chunk the 64
// bytes [hi c0-7 | lo c0-7 | hi c8-15 | lo c8-15] (fp16 halves of value * a_scale), written by cs_layernorm_pair16 and the
// pair-emitting GEMM epilogues.  Same bytes, row stride and 64-byte gather pieces as the fp32 tensor it replaces (the
// separate hi / lo images of PRE halve the piece size: measured neutral on the per-tap gather path in r2), and the two
// 16-byte pieces a lane reads are exactly its hi and lo fragments: the K loop carries no conversion VALU.
//
// TPK (r3) = taps per kd of the slab path: 9 for the 3x3x3 convs; 4 for the 3x2x2 / 2x2x2 kernels of the Upsample convs
// folded onto the source grid (cs_conv_gemm_up2: per output parity class two source taps per doubled dim, with the window
// starting at -pad where pad = 1 - parity), which ran on the per-tap gather path at 320-375 TF/s.  The slab then holds
// BM + W + 1 rows starting at (kd - pd) planes - ph lines - pw voxels from the tile's first row, the (kh, kw) tap reads it
// at shift kh * W + kw, and the weight ring has FOUR stages so that a chunk's stage is its tap index (nine taps = three
// turns of a three-stage ring; four taps = one turn of a four-stage one): every address stays a compile-time constant.
//
// PW (r3) = pointwise: 1x1x1, stride 1, no upsampling -- every token / Linear GEMM and skip convolution.  Output row m reads
// source row m, so the per-workgroup source-row tables (three integer divisions per row, an atomicMin, two barriers in the
// prologue) and the per-DMA-instruction table look-up in the K loop are replaced by one validity bit per lane.
// per-class operands of a batched folded-Upsample launch (TPK == 4 kernels; n <= 1: unused)
struct CsClsBatch {
  const void* w[8];
  const void* w_lo[8];
  float acc_scale[8];
  int n;
};

// The slices of output tile `otile` have each stored their partial tile to ws[slice][M][cout].  Publish, arrive, and -- the
// LAST R ARRIVERS only (r6) -- wait for the rest, then sum rows [share * BM/R, +BM/R) of the tile over the slices IN SLICE
// ORDER and apply the epilogue: per element the arithmetic of splitk_reduce_epi_kernel / splitk_reduce_kernel (cs_gemm.hip),
// per (16-row block, column) the same fp64 row-order sums for gn_part, the same pair conversion -- bit for bit the two-kernel
// form whichever workgroup ends up with which share.
// Visibility (MI355X_MICROARCH.md "inter-workgroup visibility", the write-through form): the partial tile goes out as 16-byte
// `sc1` stores -> `s_waitcnt vmcnt(0)` (inline asm) -> barrier -> one lane's relaxed agent-scope atomic on the tile's counter;
// the reducers poll that counter with relaxed agent-scope loads and read the partials with `sc1` loads (never served by a
// CU's L1).  No release / acquire fence on either side: the stores were written through, the loads bypass L1.
// Roles by ARRIVAL TICKET, not by slice index (r6): HIP promises nothing about dispatch order or residency, and with
// fixed roles (slices 0 .. R-1 reduce) the FIRST-dispatched workgroups of a tile were the ones that span while later slices
// might still be waiting for a CU -- safe only while the whole launch is resident.  The ticket of the atomic arrive decides
// instead: the first splits - R arrivers leave at once (their CUs are free for slices not yet dispatched), the last R arrivers
// take shares 0 .. R-1 and wait only for slices that arrived at a CU before they could finish.  A launch that is not fully
// resident (CUs held by another stream, a CU mask) therefore still completes; the bounded wait stays as a backstop and
// raises CS_STATUS_SPLITK_TIMEOUT, which the hosts answer by re-running with the two-kernel form (no_fused_reduce).
template <int NT, int BM, int BN>
__device__ __forceinline__ void fused_splitk_reduce(const CsFuseK& f, const float* __restrict__ ws, unsigned char* smem, int M,
                                                    int cout, int m0, int n0, int splits, int otile, int tid) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));
  int32_t* const arrive = f.sync + 2 * otile;
  // (the partial tile went out as 16-byte WRITE-THROUGH stores -- `sc1` -- so publishing is "drain, then arrive": no
  // buffer_wbl2 over the 229 KB this workgroup just dirtied; v1 of this seam used plain stores + a release fence and cost
  // ~23 us per conv against ~15 for the second launch, profiles/r05_a_fused_ab.txt)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                           // every wave's partial stores are drained
  int* const tick = reinterpret_cast<int*>(smem);            // (the staging LDS is free: the caller synchronised)
  if (tid == 0) *tick = __hip_atomic_fetch_add(arrive, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const int R = f.reducers;
  const int split = *tick - (splits - R);                    // this workgroup's share of the tile's rows, < 0 = none
  if (split < 0) return;                                     // (workgroup-uniform)
  if (tid == 0) {
    int spins = 0;
    while (__hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < splits) {
      __builtin_amdgcn_s_sleep(4);
      if (++spins > (1 << 23)) {                             // ~ seconds: a slice that never got a CU, not a hang
        if (f.status) atomicOr(f.status, CS_STATUS_SPLITK_TIMEOUT);
        break;
      }
    }
  }
  __syncthreads();                                           // (the partials are read with `sc1` loads: no acquire fence needed)
  constexpr int C4 = BN / 4;                                 // float4 columns of the tile
  constexpr int UNITS = 16 * C4;                             // one 16-row statistics block
  static_assert(UNITS % 2 == 0 && BN <= NT && 16 * BN * 4 <= 32768, "fused split-K reduce geometry");
  float* const lv = reinterpret_cast<float*>(smem);          // [16][BN] final values of the block (gn_part only)
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)ws, 0, (unsigned)((int64_t)splits * M * cout * 4), 0x00020000);
  const bool gstat = f.gn_part != nullptr, opair = f.out_format == 2;
  const int rows_per = BM / R;
  float oamax = 0.f;
  for (int blk = 0; blk < rows_per / 16; ++blk) {
    const int r0 = split * rows_per + 16 * blk;              // first row of the block inside the tile
    if (m0 + r0 >= M) break;                                 // (uniform; blocks past M hold nothing: rows per sample % 16 == 0)
    for (int u0 = 0; u0 < UNITS; u0 += NT) {
      const int u = u0 + tid;
      const int lrow = u / C4, c4 = u - lrow * C4;
      const int m = m0 + r0 + lrow, n = n0 + 4 * c4;
      const bool ok = u < UNITS && m < M && n < cout;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (ok) {
        // `sc1` loads (served by L2 / memory, never by this CU's L1) of what the slices stored `sc1`; the slice stride
        // M * cout * 4 bytes stays far below the descriptor's 4 GiB at every size the resident-launch rule admits
        const unsigned off0 = (unsigned)(((int64_t)m * cout + n) * 4);
        const unsigned sstride = (unsigned)((int64_t)M * cout * 4);
        v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, off0, 0, 16));
        for (int s = 1; s < splits; s += 8) {                // eight loads in flight, added in slice order
          f32x4 t[8];
#pragma unroll
          for (int q = 0; q < 8; ++q)
            if (s + q < splits)
              t[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, off0 + (unsigned)(s + q) * sstride, 0, 16));
#pragma unroll
          for (int q = 0; q < 8; ++q)
            if (s + q < splits) v += t[q];
        }
        if (f.bias) v += *reinterpret_cast<const f32x4*>(f.bias + n);
        if (f.scale) v = v * *reinterpret_cast<const f32x4*>(f.scale + n) + *reinterpret_cast<const f32x4*>(f.shift + n);
        if (f.rowvec) v += *reinterpret_cast<const f32x4*>(f.rowvec + (int64_t)(m / f.rv_rows) * f.ldrv + n);
        if (f.act != CS_ACT_NONE) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = cs_act(v[e], f.act);
        }
        if (f.res) v += *reinterpret_cast<const f32x4*>(f.res + (int64_t)m * f.ldr + n);
      }
      if (gstat && u < UNITS) *reinterpret_cast<f32x4*>(lv + lrow * BN + 4 * c4) = v;     // (masked rows / columns: zeros)
      if (opair) {                                           // uniform branch: every lane takes part in the half swap
        h4 hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float o = v[e] * f.out_scale;
          oamax = fmaxf(oamax, fabsf(o));
          hi[e] = (_Float16)o;
          lo[e] = (_Float16)(o - (float)hi[e]);
        }
        const u32x2 H = __builtin_bit_cast(u32x2, hi), L = __builtin_bit_cast(u32x2, lo);
        const bool odd = tid & 1;                            // (UNITS and NT are even: lanes 2t / 2t + 1 = columns 8g / 8g + 4 of a row)
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
          char* row = reinterpret_cast<char*>(f.out + (int64_t)m * f.ldo);
          *reinterpret_cast<u32x4*>(row + (n >> 4) * 64 + ((n & 8) ? 32 : 0) + ((n & 4) ? 16 : 0)) = r;
        }
      } else if (ok) {
        *reinterpret_cast<f32x4*>(f.out + (int64_t)m * f.ldo + n) = v;
      }
    }
    if (gstat) {
      __syncthreads();
      if (tid < BN && n0 + tid < cout) {
        double ts = 0.0, tq = 0.0;
#pragma unroll
        for (int r2 = 0; r2 < 16; ++r2) {
          const double d = (double)lv[r2 * BN + tid];
          ts += d;
          tq += d * d;
        }
        double* o = f.gn_part + ((int64_t)((m0 + r0) / 16) * f.gn_ld + n0 + tid) * 2;
        o[0] = ts;
        o[1] = tq;
      }
      __syncthreads();
    }
  }
  if (opair && f.status && oamax >= 65504.f) atomicOr(f.status, CS_STATUS_F16X3_OVERFLOW);
  // leave: the last reducer of the tile returns both counters to zero (every reducer has left its wait by then; the slices
  // that do not reduce arrived before that wait could end)
  __syncthreads();
  if (tid == 0) {
    const int d = __hip_atomic_fetch_add(arrive + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (d == R - 1) {
      __hip_atomic_store(arrive, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(arrive + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// f(integral_constant<int, I>) for I = 0 .. N-1, unrolled at compile time (the ring positions of the K loop)
template <class F, int... I>
__device__ __forceinline__ void static_steps(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}

template <int WMB, int WNB, int WAVES_M, int WAVES_N, bool PRE, int SLAB = 0, bool PAIR = false, int TPK = 9, bool PW = false>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, 2) void conv_gemm_f16x3_kernel(const CsConvGemm p, int M, int tiles_n,
                                                              int taps_hw, int kw_, int kg_per_tap,
                                                              long long x_bytes, unsigned w_bytes, int vec_epilogue,
                                                              int splits, int omap_f, int omap_p_in, const CsClsBatch cb,
                                                              const CsFuseK fz, int u_base, int u_count) {
  constexpr int BM = 32 * WMB * WAVES_M;
  constexpr int BN = 32 * WNB * WAVES_N;
  constexpr int NW = WAVES_M * WAVES_N;            // waves per workgroup (4, or 8 for the 256-row tile)
  constexpr int NT = 64 * NW;
  // ---- LDS map (bytes) ----
  static_assert(!SLAB || (WMB <= 2 && WAVES_N == 1), "slab path: one or two row blocks per wave");
  static_assert(!PAIR || (!PRE && SLAB == 0), "interleaved operand pairs: per-tap gather path only");
  static_assert(TPK == 9 || ((TPK == 4 || TPK == 3) && SLAB != 0), "taps per kd: 3x3 or (slab path only) 2x2 / 3x1");
  static_assert(!PW || SLAB == 0, "pointwise GEMMs take the gather path (there is one tap)");
  constexpr int KW_ = TPK == 9 ? 3 : TPK == 4 ? 2 : 1;   // kw extent of a slab super-chunk ...
  constexpr int KH_ = TPK / KW_;                         // ... and its kh extent (3x3, 2x2, or 3x1: the Winograd-W position GEMMs)
  // wave-instructions per slab (SLAB = widest line W): fp32 rows of 64 B, 16 per instruction; PRE: a hi and a lo image
  // of 32 B rows, 32 per instruction each
  constexpr int SLAB_ROWS = BM + (KH_ - 1) * SLAB + (KW_ - 1);   // 3x3 taps: BM + 2W + 2 rows; 2x2: BM + W + 1; 3x1: BM + 2W
  constexpr int SLAB_IMG_WI = (SLAB_ROWS + 31) / 32;
  constexpr int SLAB_WI = PRE ? 2 * SLAB_IMG_WI : (SLAB_ROWS + 15) / 16;
  constexpr int SLAB_IMG = SLAB_IMG_WI * 1024;     // PRE: byte offset of the lo image inside a slab
  constexpr int SLAB_BYTES = SLAB ? SLAB_WI * 1024 : 0;
  constexpr int RING0 = 2 * SLAB_BYTES;            // two slabs, then the ring
  constexpr int A_BYTES = SLAB ? 0 : BM * 64;      // raw fp32 [BM][16], or (PRE) fp16 hi [BM][16] + lo [BM][16]
  constexpr int B_BYTES = 2 * BN * 16;             // one fp16 image [2 k-groups][BN][8]
  constexpr int STAGE = A_BYTES + 2 * B_BYTES;
  // ring depth: the DMA stream runs PF = NSTAGE - 1 chunks ahead of the MFMAs.  Two chunks cover the latency behind
  // the big tiles' 21-MFMA chunks; the 64x64 tile (small launches: a few hundred short K loops of 3 MFMAs per chunk)
  // is pure DMA latency at that depth, and its 8 KB stages leave room for five chunks in flight
#ifdef CS_RING3      // A/B timing builds (tools/ring_ab.sh): the round-1 depth everywhere
  constexpr int NSTAGE = 3;
#else
  // (r3: the pointwise 256x224 tile -- one workgroup per CU either way -- runs a four-stage ring, three chunks of DMA in
  // flight: 448->448 117.4 -> 114.4 us, 1792->448 317.5 -> 309.2, 81.41 -> 81.14 ms/step same box; -DCS_PW_RING3 for A/B)
#ifdef CS_PW_RING3
  constexpr int NSTAGE = (BM == 64 && BN == 64) ? 6 : (TPK == 4 ? 4 : 3);
#else
#ifndef CS_RING64
#define CS_RING64 6
#endif
  constexpr int NSTAGE = (BM == 64 && BN == 64) ? CS_RING64 : ((TPK == 4 || (PW && BM == 256 && BN == 224)) ? 4 : 3);
#endif
#endif
  constexpr int PF = NSTAGE - 1;
  constexpr int DUMP = RING0 + NSTAGE * STAGE;     // 1 KB: where surplus DMA wave-instructions land
  constexpr int ROWBASE = DUMP + 1024;             // int32 [BM]: source row of the window origin
  constexpr int DELTA = ROWBASE + (SLAB ? 0 : BM * 4);   // int16 [MAX_TAPS][BM]: source row - rowbase, or INVALID
  constexpr int ROWMIN = DELTA + (SLAB ? 0 : MAX_TAPS * BM * 2);   // int32: smallest source row of the tile (descriptor window base)
  // the pipelined epilogue re-uses the LDS from offset 0 (staging + two residual slabs per wave + the vector rows)
  constexpr int EP_KU = (4 * (32 * WNB / 4) + 63) / 64;
  constexpr int EPI_NEED = NW * (16 * 32 * WNB) + NW * 2 * EP_KU * 1024 + 2048;
  // (TPK == 4, the folded Upsample convs: int32 [BM] output-row table of the scattered store, see the epilogue)
  constexpr int OTAB = ((ROWMIN + 16 > EPI_NEED) ? ROWMIN + 16 : EPI_NEED);
  constexpr int LDS_BYTES = OTAB + (TPK == 4 ? BM * 4 : 0);
  constexpr short INVALID = (short)0x8000;
  // ---- DMA schedule: wave-instructions of 64 x 16 B ----
  constexpr int A_WI = BM / 16;                    // A wave-instructions per chunk
  constexpr int B_WI = BN / 32;                    // per B image
  constexpr int A_PW = SLAB ? 0 : A_WI / NW;       // per wave (slab: the A operand is not part of the chunk stream)
  constexpr int A_PWN = A_PW > 0 ? A_PW : 1;
  constexpr int B_PW = (2 * B_WI + NW - 1) / NW;   // per wave, hi + lo together (surplus ones go to DUMP)
  constexpr int D = A_PW + B_PW;                   // DMA instructions per wave per chunk
  static_assert(A_WI % NW == 0, "A tile must split evenly over the waves");
  static_assert(D <= 7, "vmcnt immediates");
  __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
  // Class-dependent operands.  TPK == 4 only: ONE launch may cover all output parity classes of a folded Upsample conv
  // (cs_conv_gemm_up2, cb.n = 4 or 8) -- the virtual tile range is then [class][tile], each class with its own packed
  // weights / accumulator scale, its pads = 1 - parity in the doubled dims and its scatter parity; at 32 objects the 4^3
  // level's classes are 192 workgroups each (a quarter of the CUs idle per launch), together 768 = three even rounds.
  const void* w_hi_ = p.w;
  const void* w_lo_ = p.w_lo;
  float acc_scale_ = p.acc_scale;
  int pad_d = p.pd, pad_h = p.ph, pad_w = p.pw, omap_p = omap_p_in;
  int cls_id = 0;                                                 // parity class of a batched folded-Upsample launch
  if constexpr (TPK == 4) {
    if (cb.n > 1) {
      // (r5: omap_f bit 3 = the K-SLICED class batch of small launches -- plain partial tiles [class][slice][M][cout] to the
      // workspace, summed and scattered by up2_reduce_scatter_kernel, cs_gemm.hip -- instead of the direct scattered store)
      const int per_cls = ((M + BM - 1) / BM) * tiles_n * splits;
      const int cls = tile / per_cls;
      cls_id = cls;
      tile -= cls * per_cls;
      w_hi_ = cb.w[cls];
      w_lo_ = cb.w_lo[cls];
      acc_scale_ = cb.acc_scale[cls];
      const int nw = 1 + (omap_f & 1), nh = 1 + ((omap_f >> 1) & 1);
      const int qw = cls % nw, qh = (cls / nw) % nh, qd = cls / (nw * nh);
      pad_d = (omap_f & 4) ? 1 - qd : 1;
      pad_h = (omap_f & 2) ? 1 - qh : 1;
      pad_w = (omap_f & 1) ? 1 - qw : 1;
      omap_p = (qd << 2) | (qh << 1) | qw;
    }
  }
  // split-K: consecutive virtual tiles are the K slices of one output tile (same XCD: they share the A rows).
  // Slab kernel with K slices (the large-batch four-way cut of the 4^3-level convs: 49 MB of weights per conv, far
  // beyond an XCD's L2): row tiles fastest instead, so the workgroups an XCD runs together stream the SAME weight
  // slice (one or two (column tile, K slice) pairs per XCD instead of twelve).
  // r6: a launch of the Winograd-W position GEMMs may cover a RANGE of (position, column tile) UNITS [u_base, u_base + u_count)
  // (u_count = 0: everything, the plain decode below) -- unit u = position u / tiles_n, column tile u % tiles_n, each with the
  // R = tiles_m / positions row tiles of its position.  A launch whose tile count is not a whole number of rounds of the chip
  // runs as a main launch over whole rounds (unsliced) and a K-sliced tail launch over the remaining units
  // (cs_gemm.hip::conv_wino).  Units, not row tiles: how a sum is cut then depends on (position, column) only, never on the
  // row -- a sample's result stays independent of its place in the batch.
  int split, tn, tm;
  if (TPK == 3 && u_count > 0) {
    const int R = ((M + BM - 1) / BM) / (omap_p_in > 1 ? omap_p_in : 1);
    const int r = tile % R;                      // row tiles fastest: neighbours stream the same weight slice
    tile /= R;
    split = tile % splits;
    const int u = u_base + tile / splits;
    tn = u % tiles_n;
    tm = (u / tiles_n) * R + r;
  } else if (SLAB != 0 && splits > 1) {
    const int tiles_m = (M + BM - 1) / BM;
    tm = tile % tiles_m;
    tile /= tiles_m;
    split = tile % splits;
    tn = tile / splits;
  } else {
    split = tile % splits;
    tile /= splits;
    tn = tile % tiles_n;
    tm = tile / tiles_n;
  }
  const int m0 = tm * BM;
  const int n0 = tn * BN;
  // TPK == 3 (r5): the four Winograd-W position GEMMs of one 3x3x3 conv in ONE launch -- their transformed operands are
  // stacked along the batch ([4][nb][D][H][W/2][C]: rows of class q = [q * M / 4, (q + 1) * M / 4), whole row tiles), their
  // transformed weights are four consecutive packed images; everything else (pads 1 / 1 / 0, plain stores) is class-blind.
  if constexpr (TPK == 3) {
    if (omap_p_in > 1) {
      const int tiles_m = (M + BM - 1) / BM;
      const int q = tm / (tiles_m / omap_p_in);
      w_hi_ = reinterpret_cast<const char*>(p.w) + (size_t)q * w_bytes;
      w_lo_ = reinterpret_cast<const char*>(p.w_lo) + (size_t)q * w_bytes;
    }
  }

  const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc((void*)w_hi_, 0, w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t lrs = __builtin_amdgcn_make_buffer_rsrc((void*)w_lo_, 0, w_bytes, 0x00020000);

  // ---- source-row tables, built once per workgroup ----
  const int ntaps = p.kd * taps_hw;
  int* rowbase = reinterpret_cast<int*>(smem + ROWBASE);
  int* rowmin = reinterpret_cast<int*>(smem + ROWMIN);
  short* delta = reinterpret_cast<short*>(smem + DELTA);
  if constexpr (!SLAB && !PW) {
  if (tid == 0) *rowmin = 0x7fffffff;
  __syncthreads();
  {
    // NT / BM threads per output row: the row's (n, od, oh, ow) decomposition -- three integer divisions -- is done
    // once, the taps are then walked with adds and compares only
    const int vdin = p.din << p.ud, vhin = p.hin << p.uh, vwin = p.win << p.uw;
    constexpr int TPR = NT / BM >= 1 ? NT / BM : 1;       // threads sharing a row (taps are dealt round-robin)
    for (int row = tid / TPR; row < BM; row += NT / TPR) {
      const int sub = tid % TPR;
      const int m = m0 + row;
      int mm = m < M ? m : 0;
      const int ow = mm % p.wout;
      mm /= p.wout;
      const int oh = mm % p.hout;
      mm /= p.hout;
      const int od = mm % p.dout;
      const int n = mm / p.dout;
      // reference position: the (clamped) source voxel of the window origin
      const int cd = min(max(od * p.sd - pad_d, 0), vdin - 1) >> p.ud;
      const int ch = min(max(oh * p.sh - pad_h, 0), vhin - 1) >> p.uh;
      const int cw = min(max(ow * p.sw - pad_w, 0), vwin - 1) >> p.uw;
      const int base = ((n * p.din + cd) * p.hin + ch) * p.win + cw;
      if (sub == 0) {
        rowbase[row] = base;
        if (m < M) atomicMin(rowmin, base);      // not simply row 0: with upsampling / clamped borders a later row of
      }                                          // a tile that starts mid-line can sit at a smaller source row
      const int vd0 = od * p.sd - pad_d, vh0 = oh * p.sh - pad_h, vw0 = ow * p.sw - pad_w;
      int t = 0;
      for (int kd_ = 0; kd_ < p.kd; ++kd_)
        for (int kh_ = 0; kh_ < p.kh; ++kh_)
          for (int kwi = 0; kwi < kw_; ++kwi, ++t) {
            if (t % TPR != sub) continue;
            short dl = INVALID;
            const int vd = vd0 + kd_, vh = vh0 + kh_, vw = vw0 + kwi;
            if (m < M && (unsigned)vd < (unsigned)vdin && (unsigned)vh < (unsigned)vhin &&
                (unsigned)vw < (unsigned)vwin) {
              const int r = ((n * p.din + (vd >> p.ud)) * p.hin + (vh >> p.uh)) * p.win + (vw >> p.uw);
              dl = (short)(r - base);
            }
            delta[t * BM + row] = dl;
          }
    }
  }
  __syncthreads();
  }

  // The activation buffer descriptors are per workgroup: based at the smallest source row this tile can touch (every
  // valid tap sits at or after its row's clamped window origin, so that is the minimum of rowbase over the tile's
  // rows), with 32-bit offsets inside a window of a few MB.  The tensor itself may therefore be larger than
  // the 4 GiB a single descriptor spans (288 GB of HBM: 200+ objects per batch at the 16^3 x 672-channel level).
  // (slab: the lowest row any of the three kd slabs can start at)
  const int row_lo = SLAB ? max(0, m0 - pad_d * p.hin * p.win - pad_h * p.win - pad_w)
                     : PW ? m0 : __builtin_amdgcn_readfirstlane(*rowmin);
  const long long x_skip = (long long)row_lo * p.lda * (PRE ? 2 : 4);
  const long long x_left = x_bytes - x_skip;
  const unsigned x_win = x_left > 0xFFE00000LL ? 0xFFE00000u : (unsigned)x_left;
  const __amdgpu_buffer_rsrc_t xrs =
      __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.x + x_skip), 0, x_win, 0x00020000);
  const __amdgpu_buffer_rsrc_t xlrs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)((const char*)(PRE ? p.x_lo : (const void*)p.x) + x_skip), 0, x_win, 0x00020000);

  // r4: operand scale from a bound on the tensor's magnitude (CsConvGemm.a_bound, left by the GroupNorm finalize kernels):
  // the largest power of two that keeps the bound inside the fp16 range; acc_scale_ follows by the exact ratio.  Uniform.
  float a_scale = p.a_scale;
  if constexpr (!PRE && !PAIR) {
    if (p.a_bound) {
      const float m = *p.a_bound;
      int ex = 0;
      float s2 = 1.0995116e12f;                              // 2^40: an all-zero (or empty) tensor
      if (m > 0.f && m < 3.0e38f) {
        (void)frexpf(65000.0f / m, &ex);
        s2 = ldexpf(1.0f, min(max(ex - 1, -8), 40));
      }
      acc_scale_ *= a_scale / s2;
      a_scale = s2;
    }
  }
  const int chunks_per_tap = kg_per_tap >> 1;   // cin16 / 16
  const int nk_all = ntaps * chunks_per_tap;
  // this workgroup's slice of the chunk sequence (slab path: whole super-chunks of nine taps)
  constexpr int KGRAN = SLAB ? TPK : 1;
  const int per_split = ((nk_all / KGRAN + splits - 1) / splits) * KGRAN;
  const int k_first = split * per_split;
#if CS_ABLATE & 2048
  const int nk = 0;
#else
  const int nk = max(0, min(nk_all, k_first + per_split) - k_first);
#endif

  // ---- per-lane DMA constants ----
  // A wave-instruction w (0..A_WI-1) covers units 64w..64w+63: row = 16w + lane/4, LDS slot q = lane&3 holds
  // global 16-byte piece q ^ ((row>>2)&3).
  int a_rowbase[A_PWN];
  unsigned a_piece[A_PWN];
  int a_rowidx[A_PWN];
#pragma unroll
  for (int i = 0; i < A_PW; ++i) {
    const int w = wave * A_PW + i;
    if constexpr (PRE) {
      // image w / (A_WI/2) (0 = hi, 1 = lo); rows of 16 halves = two 16-byte pieces, slot q holds piece q ^ ((row>>3)&1)
      const int within = w % (A_WI / 2);
      const int row = 32 * within + (lane >> 1);
      a_rowidx[i] = row;
      if constexpr (PW) a_rowbase[i] = (m0 + row < M) ? row : -1;       // pointwise: source row = output row; -1 = past M
      else a_rowbase[i] = rowbase[row] - row_lo;
      a_piece[i] = (unsigned)(((lane & 1) ^ ((row >> 3) & 1)) * 8);
    } else {
      const int row = 16 * w + (lane >> 2);
      a_rowidx[i] = row;
      if constexpr (PW) a_rowbase[i] = (m0 + row < M) ? row : -1;
      else a_rowbase[i] = rowbase[row] - row_lo;
      a_piece[i] = (unsigned)(((lane & 3) ^ ((row >> 2) & 3)) * 4);   // first channel of the piece
    }
  }
  // B wave-instruction v (0..2*B_WI-1): image v / B_WI (0 = hi, 1 = lo), units 64*(v % B_WI) .. +63
  unsigned b_off[B_PW];     // byte offset inside the chunk's [2][cout][8] slab, or OOB
#pragma unroll
  for (int i = 0; i < B_PW; ++i) {
    const int v = wave * B_PW + i;
    const int img = v / B_WI;
    const int u = (v - img * B_WI) * 64 + lane;           // unit inside the image: kg = u / BN, n = u % BN
    const int kg = u / BN;
    const int n = u - kg * BN;
    b_off[i] = (v < 2 * B_WI && n0 + n < p.cout) ? (unsigned)((kg * p.cout + n0 + n) * 16) : OOB;
  }

  auto issue_dma = [&](int tap, int cc, int stage) {
    unsigned char* st = smem + RING0 + stage * STAGE;
    const bool skip_a = ((CS_ABLATE & 128) && tap % 3 != 0) || ((CS_ABLATE & 256) && tap % 9 != 0);   // slab what-if
#pragma unroll
    for (int i = 0; i < A_PW; ++i) {
      if (skip_a) break;
      const int c = cc * BKH + (int)a_piece[i];
      constexpr unsigned ESZ = PRE ? 2u : 4u;
      unsigned off;
      if constexpr (PW) {
        off = (a_rowbase[i] >= 0 && c < p.cin) ? (unsigned)a_rowbase[i] * ((unsigned)p.lda * ESZ) + (unsigned)c * ESZ : OOB;
      } else {
        const short dl = delta[tap * BM + a_rowidx[i]];
        off = (dl != INVALID && c < p.cin)
                  ? (unsigned)(a_rowbase[i] + (int)dl) * ((unsigned)p.lda * ESZ) + (unsigned)c * ESZ
                  : OOB;
      }
      if (CS_ABLATE & 2) off = OOB;
      if ((CS_ABLATE & 64) && off != OOB) off &= 0x3FF0u;      // every fetch from one 16 KB window (cache hits)
      const int w = wave * A_PW + i;                      // wave-uniform
      if constexpr (PRE) {
        if (w >= A_WI / 2)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(xlrs, st + w * 1024, 16, off, 0, 0, 0);
        else
          __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, st + w * 1024, 16, off, 0, 0, 0);
      } else {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, st + w * 1024, 16, off, 0, 0, 0);
      }
    }
    const unsigned kbase = (unsigned)((tap * kg_per_tap + cc * 2) * p.cout) * 16u;
#pragma unroll
    for (int i = 0; i < B_PW; ++i) {
      const int v = wave * B_PW + i;                       // wave-uniform
      const int img = v / B_WI;
      unsigned off = (b_off[i] == OOB || cc >= chunks_per_tap) ? OOB : b_off[i] + kbase;
      if (CS_ABLATE & 2) off = OOB;
      if ((CS_ABLATE & 64) && off != OOB) off &= 0x3FF0u;
      unsigned char* dst = (v < 2 * B_WI) ? st + A_BYTES + img * B_BYTES + (v - img * B_WI) * 1024 : smem + DUMP;
      if (img == 1)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(lrs, dst, 16, off, 0, 0, 0);
      else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(hrs, dst, 16, off, 0, 0, 0);
    }
  };

  f32x16 acc[WMB][WNB];
#pragma unroll
  for (int i = 0; i < WMB; ++i)
#pragma unroll
    for (int j = 0; j < WNB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // DMA stream position (chunk index q -> tap = q % ntaps, cc = q / ntaps), two chunks ahead of compute
  int dtap = k_first % ntaps, dcc = k_first / ntaps;
  auto advance = [&]() {
    if (++dtap == ntaps) {
      dtap = 0;
      ++dcc;
    }
  };
#pragma unroll
  for (int q = 0; q < PF; ++q) {
    issue_dma(dtap, dcc, q);
    advance();
  }

  // fragment addressing
  int a_frag[WMB][2];
#pragma unroll
  for (int i = 0; i < WMB; ++i) {
    const int row = wm0 + 32 * i + l31;
    if constexpr (PRE) {
      a_frag[i][0] = row * 32 + ((half ^ ((row >> 3) & 1)) * 16);          // hi image
      a_frag[i][1] = a_frag[i][0] + BM * 32;                                // lo image
    } else {
      const int s = (row >> 2) & 3;
      a_frag[i][0] = row * 64 + (((2 * half) ^ s) * 16);
      a_frag[i][1] = row * 64 + (((2 * half + 1) ^ s) * 16);
    }
  }
  const int b_frag = A_BYTES + (half * BN + wn0 + l31) * 16;

  // Software pipeline: the hi/lo split of chunk k+1's A fragment runs inside chunk k's MFMA stream.
  //   top of iteration k : outstanding DMAs (oldest first) = A(k+1) B(k+1) ... A(k+PF-1) B(k+PF-1)
  //   wait vmcnt((PF-2)*D + B_PW) : everything up to A(k+1) has landed, B(k+1) and the later chunks may still fly
  //   s_barrier          : ... for every wave; every wave has also left iteration k-1
  //   issue A(k+PF), B(k+PF) into the stage iteration k-1 vacated (A first, so the next wait covers it)
  //   MFMAs on B(k) with the already-split A(k)  ||  read + split A(k+1)
  float amax = 0.f;
  auto load_a = [&](int st, h8 (&hi)[WMB], h8 (&lo)[WMB]) {
    const unsigned char* s = smem + RING0 + st * STAGE;
#pragma unroll
    for (int i = 0; i < WMB; ++i) {
      if constexpr (PRE) {
        hi[i] = *reinterpret_cast<const h8*>(s + a_frag[i][0]);
        lo[i] = *reinterpret_cast<const h8*>(s + a_frag[i][1]);
      } else {
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(s + a_frag[i][0]);
        const f32x4 x1 = *reinterpret_cast<const f32x4*>(s + a_frag[i][1]);
#if CS_ABLATE & 512      // what-if "activations arrive split" (timing only)
        hi[i] = __builtin_bit_cast(h8, x0);
        lo[i] = __builtin_bit_cast(h8, x1);
#else
        if constexpr (PAIR) {          // the producer already split: the two pieces ARE the hi / lo fragments
          hi[i] = __builtin_bit_cast(h8, x0);
          lo[i] = __builtin_bit_cast(h8, x1);
        } else {
          split8(x0, x1, a_scale, hi[i], lo[i], amax);
        }
#endif
      }
    }
  };
  // ---- slab path: per-lane validity mask, slab DMA, shifted fragment reads ----
  constexpr int SLAB_PW = (SLAB_WI + NW - 1) / NW;        // slab wave-instructions per wave (every wave issues all of
                                                          // them, surplus ones as zero-fills: the counts below are exact)
  const int s_w = p.win, s_hw = p.hin * p.win;
  const int s_rows = p.nb * p.din * s_hw;                  // source rows in the tensor
  const int s_need = BM + (KH_ - 1) * s_w + (KW_ - 1);     // slab rows this problem uses
  // (per row block i of the wave -- one for the 256-row tiles, two for the 512-row ones)
  unsigned vmask[WMB];                                     // bit tap: that tap of this lane's output voxel is inside the volume
  int sl_row[SLAB_PW];
  unsigned sl_piece[SLAB_PW];
  int sl_a0[WMB][TPK];                                     // fragment byte offset inside the slab, per (kh, kw)
  if constexpr (SLAB != 0) {
#pragma unroll
    for (int i = 0; i < WMB; ++i) {
      const int rowl = wm0 + 32 * i + l31;                 // the lane's fragment row inside the tile
#pragma unroll
      for (int t9 = 0; t9 < TPK; ++t9) {
        const int prow = rowl + (t9 / KW_) * s_w + t9 % KW_;
        if constexpr (PRE)
          sl_a0[i][t9] = prow * 32 + (((half ^ (prow >> 3)) & 1) << 4);      // same swizzle as the ring's fp16 images
        else
          sl_a0[i][t9] = prow * 64 + ((((2 * half) ^ (prow >> 2)) & 3) << 4);
      }
      vmask[i] = 0;
      const int m = m0 + rowl;
      if (m < M) {
        int mm = m;
        const int ow = mm % p.win;
        mm /= p.win;
        const int oh = mm % p.hin;
        mm /= p.hin;
        const int od = mm % p.din;
#pragma unroll
        for (int t = 0; t < 3 * TPK; ++t) {               // (kd >= p.kd: never used)
          const int kd_ = t / TPK, kh_ = (t % TPK) / KW_, kwi = t % KW_;
          if ((unsigned)(od + kd_ - pad_d) < (unsigned)p.din && (unsigned)(oh + kh_ - pad_h) < (unsigned)p.hin &&
              (unsigned)(ow + kwi - pad_w) < (unsigned)p.win)
            vmask[i] |= 1u << t;
        }
      }
    }
    if constexpr (PRE) {
      if (tid < 4) reinterpret_cast<int*>(smem + ROWMIN)[tid] = 0;     // 16 zero bytes: what a masked-out tap reads
      __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < SLAB_PW; ++i) {
      if constexpr (PRE) {
        const int j = 32 * ((wave + NW * i) % SLAB_IMG_WI) + (lane >> 1);
        sl_row[i] = j;
        sl_piece[i] = (unsigned)(((lane & 1) ^ ((j >> 3) & 1)) * 8);
        continue;
      }
      const int j = 16 * (wave + NW * i) + (lane >> 2);   // slab row of this lane in its i-th instruction
      sl_row[i] = j;
      sl_piece[i] = (unsigned)(((lane & 3) ^ ((j >> 2) & 3)) * 4);
    }
  }
  // slab of super-chunk sc (= channel chunk sc / 3, kd = sc % 3): source rows m0 + (kd-1)*H*W - W - 1 ... in order
  const int KD = p.kd;                                     // 3, or 2 for a folded depth dimension
  auto issue_slab = [&](int sc) {
    const int cc = sc / KD, kd_ = sc - KD * cc;
    const int src0 = m0 + (kd_ - pad_d) * s_hw - pad_h * s_w - pad_w;
    unsigned char* dst = smem + (sc & 1) * SLAB_BYTES;
#pragma unroll
    for (int i = 0; i < SLAB_PW; ++i) {
      const int q = wave + NW * i;                          // wave-uniform
      const int src = src0 + sl_row[i];
      const int c = cc * BKH + (int)sl_piece[i];
      constexpr unsigned ESZ = PRE ? 2u : 4u;
      const unsigned off = (q < SLAB_WI && sl_row[i] < s_need && src >= 0 && src < s_rows && c < p.cin && cc < chunks_per_tap)
                               ? (unsigned)(src - row_lo) * ((unsigned)p.lda * ESZ) + (unsigned)c * ESZ
                               : OOB;
      // (through a variable: a conditional expression as the LDS argument makes the HOST pass drop the kernel's stub
      // without a diagnostic)
      unsigned char* d2 = q < SLAB_WI ? dst + q * 1024 : smem + DUMP;
      if (PRE && q >= SLAB_IMG_WI)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xlrs, d2, 16, off, 0, 0, 0);        // lo image
      else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, d2, 16, off, 0, 0, 0);
    }
  };
  // fragment of the chunk at (kh, kw) = T9 of the super-chunk whose slab sits in buffer `buf`: the lane's row shifted by
  // kh*W + kw, zeroed (scale 0, legacy multiply) if that tap falls outside the volume for this output voxel
  // (m9 = the kd's nine mask bits)
  auto load_a_slab = [&](auto t9_c, int i, int buf, unsigned m9, h8& hi, h8& lo) {
    constexpr int t9 = decltype(t9_c)::value;
    if constexpr (PRE) {
      // already split by the producer: two 16-byte reads; a masked-out tap reads the zero block instead
      const bool ok = (m9 >> t9) & 1u;
      const int a = buf * SLAB_BYTES + sl_a0[i][t9];
      hi = *reinterpret_cast<const h8*>(smem + (ok ? a : ROWMIN));
      lo = *reinterpret_cast<const h8*>(smem + (ok ? a + SLAB_IMG : ROWMIN));
      return;
    }
    const unsigned char* sb = smem + buf * SLAB_BYTES;
    const f32x4 x0 = *reinterpret_cast<const f32x4*>(sb + sl_a0[i][t9]);
    const f32x4 x1 = *reinterpret_cast<const f32x4*>(sb + (sl_a0[i][t9] ^ 16));
#if CS_ABLATE & 512      // what-if "activations arrive split": no conversion VALU (timing only)
    hi = __builtin_bit_cast(h8, x0);
    lo = __builtin_bit_cast(h8, x1);
    (void)m9;
#else
    cs16::split8_masked(x0, x1, ((m9 >> t9) & 1u) ? a_scale : 0.f, hi, lo, amax);
#endif
  };

  h8 ah[WMB], al[WMB];
  if constexpr (SLAB != 0) {
    // the prologue above issued B(0), B(1); the first slab goes out now and is the youngest: wait for everything
    const int sc0 = k_first / TPK;
    issue_slab(sc0);
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < WMB; ++i)
      load_a_slab(std::integral_constant<int, 0>{}, i, sc0 & 1, (vmask[i] >> (TPK * (sc0 % KD))) & ((1u << TPK) - 1u), ah[i],
                  al[i]);
  } else {
  wait_vmcnt<(PF - 1) * D>();          // chunk 0 (issued first) has landed for this wave
  __builtin_amdgcn_s_barrier();
  load_a(0, ah, al);
  }

  // One ring position per call, with the stage a compile-time constant: every LDS address (fragment reads and
  // the DMA destinations that go through M0) folds to an immediate instead of per-iteration scalar arithmetic.
  auto step = [&](auto stage_c) {
    constexpr int stage = decltype(stage_c)::value;
    constexpr int nstage = (stage + 1) % NSTAGE;
    constexpr int dstage = (stage + PF) % NSTAGE;
    if (!(CS_ABLATE & 4)) wait_vmcnt<(PF - 2) * D + B_PW>();
    if (!(CS_ABLATE & 8)) __builtin_amdgcn_s_barrier();
    const unsigned char* s = smem + RING0 + stage * STAGE;
    h8 ah2[WMB], al2[WMB];
#if CS_ABLATE & 16
    __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
    for (int j = 0; j < WNB; ++j) {
      const h8 bh = *reinterpret_cast<const h8*>(s + b_frag + j * 512);
      const h8 bl = *reinterpret_cast<const h8*>(s + b_frag + B_BYTES + j * 512);
#pragma unroll
      for (int i = 0; i < WMB; ++i) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh, acc[i][j], 0, 0, 0);
      }
      if (j == 0) {                    // behind the first MFMAs: launch the prefetch, then split A(k+1)
        if (!(CS_ABLATE & 1)) issue_dma(dtap, dcc, dstage);
        advance();
        load_a(nstage, ah2, al2);
      }
    }
#if CS_ABLATE & 16
    __builtin_amdgcn_s_setprio(0);
#endif
#pragma unroll
    for (int i = 0; i < WMB; ++i) {
      ah[i] = ah2[i];
      al[i] = al2[i];
    }
  };
  // Slab path: one super-chunk (a kd and a 16-channel chunk: nine taps) per loop iteration, fully unrolled -- which tap a
  // step computes, whether it sends the next slab, and every vmcnt are compile-time, so each step stays one
  // straight-line block the scheduler can interleave (a first version with run-time tap bookkeeping and a switch
  // over the wait count gained 3 % over the gather kernel, this one 8.5 %: DESIGN 4.4).
  //   iteration k (tap T9 of super-chunk sc), after its barrier: [T9 == 0: slab(sc + 1)]  B(k + 2)
  //   top of iteration k: only what iteration k - 1 issued may still fly
  auto sstep = [&](auto stage_c, auto t9_c, int sc, const unsigned (&m9_this)[WMB], const unsigned (&m9_next)[WMB]) {
    constexpr int stage = decltype(stage_c)::value;
    constexpr int t9 = decltype(t9_c)::value;
    constexpr int dstage = (stage + PF) % NSTAGE;
    // top of iteration k: B(k) -- issued PF iterations ago -- must have landed; what the PF - 1 iterations in between
    // issued (their weight chunks, and the slab if one of them was a tap-0 iteration) may still fly
    if (!(CS_ABLATE & 4)) wait_vmcnt<((t9 >= 1 && t9 <= PF - 1) ? SLAB_PW : 0) + (PF - 1) * B_PW>();
    if (!(CS_ABLATE & 8)) __builtin_amdgcn_s_barrier();
    const unsigned char* s = smem + RING0 + stage * STAGE;
    h8 ah2[WMB], al2[WMB];
    h8 bh, bl;
#pragma unroll
    for (int j = 0; j < WNB; ++j) {
      // (CS_ABLATE 4096 / 8192, timing only: every second / only the first B fragment pair is read from LDS)
      if (!(((CS_ABLATE & 4096) && (j & 1)) || ((CS_ABLATE & 8192) && j > 0))) {
        bh = *reinterpret_cast<const h8*>(s + b_frag + j * 512);
        bl = *reinterpret_cast<const h8*>(s + b_frag + B_BYTES + j * 512);
      }
#pragma unroll
      for (int i = 0; i < WMB; ++i) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh, acc[i][j], 0, 0, 0);
      }
      if (j == 0) {
        if constexpr (t9 == 0) issue_slab(sc + 1);       // its buffer was last read while chunk k-2 was computed
        issue_dma(dtap, dcc, dstage);
        advance();
#pragma unroll
        for (int i = 0; i < WMB; ++i) {
          if constexpr (t9 == TPK - 1)
            load_a_slab(std::integral_constant<int, 0>{}, i, (sc + 1) & 1, m9_next[i], ah2[i], al2[i]);
          else
            load_a_slab(std::integral_constant<int, t9 + 1>{}, i, sc & 1, m9_this[i], ah2[i], al2[i]);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < WMB; ++i) {
      ah[i] = ah2[i];
      al[i] = al2[i];
    }
  };
  if constexpr (SLAB != 0) {
    static_assert(((TPK == 9 || TPK == 3) && NSTAGE == 3) || (TPK == 4 && NSTAGE == 4), "a chunk's ring stage is its tap index mod NSTAGE");
    const int sc_first = k_first / TPK, sc_end = sc_first + nk / TPK;
    int kdc = sc_first % KD;                                // kd of the super-chunk
    constexpr unsigned TMASK = (1u << TPK) - 1u;
    for (int sc = sc_first; sc < sc_end; ++sc) {
      const int kdn = kdc == KD - 1 ? 0 : kdc + 1;
      unsigned m9[WMB], m9n[WMB];
#pragma unroll
      for (int i = 0; i < WMB; ++i) {
        m9[i] = (vmask[i] >> (TPK * kdc)) & TMASK;
        m9n[i] = (vmask[i] >> (TPK * kdn)) & TMASK;
      }
      if constexpr (TPK == 9) {
        sstep(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, sc, m9, m9n);
        sstep(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, sc, m9, m9n);
        sstep(std::integral_constant<int, 2>{}, std::integral_constant<int, 2>{}, sc, m9, m9n);
        sstep(std::integral_constant<int, 0>{}, std::integral_constant<int, 3>{}, sc, m9, m9n);
        sstep(std::integral_constant<int, 1>{}, std::integral_constant<int, 4>{}, sc, m9, m9n);
        sstep(std::integral_constant<int, 2>{}, std::integral_constant<int, 5>{}, sc, m9, m9n);
        sstep(std::integral_constant<int, 0>{}, std::integral_constant<int, 6>{}, sc, m9, m9n);
        sstep(std::integral_constant<int, 1>{}, std::integral_constant<int, 7>{}, sc, m9, m9n);
        sstep(std::integral_constant<int, 2>{}, std::integral_constant<int, 8>{}, sc, m9, m9n);
      } else if constexpr (TPK == 3) {
        sstep(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, sc, m9, m9n);
        sstep(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, sc, m9, m9n);
        sstep(std::integral_constant<int, 2>{}, std::integral_constant<int, 2>{}, sc, m9, m9n);
      } else {
        sstep(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, sc, m9, m9n);
        sstep(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, sc, m9, m9n);
        sstep(std::integral_constant<int, 2>{}, std::integral_constant<int, 2>{}, sc, m9, m9n);
        sstep(std::integral_constant<int, 3>{}, std::integral_constant<int, 3>{}, sc, m9, m9n);
      }
      kdc = kdn;
    }
  } else
  for (int kc = 0; kc < nk; kc += NSTAGE) {
    step(std::integral_constant<int, 0>{});
    static_steps(std::make_integer_sequence<int, NSTAGE - 1>{}, [&](auto s1) {
      constexpr int st = decltype(s1)::value + 1;
      if (kc + st < nk) step(std::integral_constant<int, st>{});
    });
  }
  wait_vmcnt<0>();   // drain the prefetches issued past the end before LDS is released
  if constexpr (!PRE && !PAIR) {
    // an activation at or beyond the fp16 range became +-inf in its hi half: tell the host (sticky flag)
    if (p.status && amax >= 65504.f) atomicOr(p.status, CS_STATUS_F16X3_OVERFLOW);
  }

#if CS_ABLATE & 1024      // what-if "no epilogue": keep the accumulators live behind a never-true store
  {
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < WMB; ++i)
#pragma unroll
      for (int j = 0; j < WNB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
    if (M == -12345) p.out[tid] = sum;
    return;
  }
#endif
  // ---- epilogue (identical contract to the fp32 kernel, after undoing the operand scales) ----
  // Fast path: the C/D layout gives a lane one column and 16 scattered rows, i.e. 112 dword stores (+112 dword
  // residual loads) per lane -- issue-bound, and for the short-K token GEMMs as long as the main loop.  Stage
  // 16 rows x (32*WNB) columns per wave through the (now idle) LDS ring and write whole rows as float4.
  // split-K: slice s owns rows [s*M, (s+1)*M) of the ws (class-batched: [class][slice])
  float* const outp = p.out + ((int64_t)cls_id * splits + split) * M * p.ldo;
  // r3: scattered store of one output parity class of a folded Upsample conv (cs_conv_gemm_up2) -- GEMM row m = source
  // voxel (n, d, h, w) lands on row ((n*Do + d*fd + pd)*Ho + h*fh + ph)*Wo + w*fw + pw of the DOUBLED grid, so the
  // classes write the final tensor themselves and the scratch tensor + interleave pass are gone.  omap_f / omap_p: bit
  // 2 / 1 / 0 = D / H / W doubled / parity.  The row map of the tile goes to LDS once (otab, relative to the tile's
  // first output row `ob`); every store path below takes its row from it.
  const bool omap = (TPK == 4) && (omap_f & 7) != 0 && !(omap_f & 8);
  int* const otab = reinterpret_cast<int*>(smem + OTAB);
  long long ob = m0, Mo = M;
  int ospan = BM;                                              // rows the tile's stores span (32-bit offset window)
  if constexpr (TPK == 4) {
    if (omap) {
      const int fd = 1 + ((omap_f >> 2) & 1), fh = 1 + ((omap_f >> 1) & 1), fw = 1 + (omap_f & 1);
      const int qd = (omap_p >> 2) & 1, qh = (omap_p >> 1) & 1, qw = omap_p & 1;
      auto orow = [&](int m) -> long long {
        const int w = m % p.win;
        int t = m / p.win;
        const int h = t % p.hin;
        t /= p.hin;
        const int d = t % p.din;
        const int n = t / p.din;
        return (((long long)n * (p.din * fd) + d * fd + qd) * (p.hin * fh) + h * fh + qh) * (p.win * fw) + w * fw + qw;
      };
      ob = orow(m0);
      Mo = (long long)M * (fd * fh * fw);
      const int mlast = min(m0 + BM, M) - 1;
      ospan = (int)(orow(mlast) - ob) + 1;
      for (int r = tid; r < BM; r += NT) otab[r] = (m0 + r < M) ? (int)(orow(m0 + r) - ob) : 0;
      __syncthreads();
    }
  }
  auto orel = [&](int row) -> int { return omap ? otab[row] : row; };     // row of the tile -> row offset from `ob`
  constexpr int WCOLS = 32 * WNB;
  constexpr int PASS_R = (NW * 16 * WCOLS * 4 <= DUMP) ? 8 : 4;             // accumulator registers per pass
  constexpr int EPI_ROWS = 2 * PASS_R;                                       // rows staged per wave per pass
  constexpr int EPI_BYTES = EPI_ROWS * WCOLS * 4;                            // per wave, per pass
  static_assert(NW * EPI_BYTES <= DUMP, "epilogue staging must fit in the ring");
  // ---- pipelined epilogue ----------------------------------------------------------------------------------------
  // The epilogue used to fetch its per-element operands with one dependent global load per float4 of output: for a
  // tile with a residual, 28 serialized HBM latencies per wave (~30 us per 256x224 tile, as long as a 448-channel
  // GEMM's whole K loop; the 448->448 vs 448->1344 rows of profiles/r02_c_gemm_table_before_pingpong.txt), and as many
  // L2 latencies for the bias and the row vector.  Here the tile's bias row and (one sample per tile) row-vector row
  // are LDS-DMA'd once, and the residual rows of pass q+1 are DMA'd into a second slab while pass q is staged,
  // combined and stored.  Stores are buffer stores whose inactive lanes carry an out-of-range offset, so every pass
  // issues exactly KU of them and the counted vmcnt below is exact.  The fused GEGLU gate takes the same route.
  {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    constexpr int QR = 2;                                    // accumulator registers per pass -> 4 rows per wave
    constexpr int NQ = 16 / QR;
    constexpr int UPR = WCOLS / 4;                           // float4 units per row
    constexpr int UN = 2 * QR * UPR;                         // units per pass per wave
    constexpr int KU = (UN + 63) / 64;                       // per lane
    constexpr int EPB = 2 * QR * WCOLS * 4;                  // staging bytes per wave
    constexpr int RSLAB = KU * 1024;                         // one residual slab per wave
    constexpr int RES0 = NW * EPB;
    constexpr int VEC0 = RES0 + NW * 2 * RSLAB;
    constexpr int BNV = (BN + 63) / 64;                      // waves that fetch the bias / row-vector rows
    static_assert(VEC0 + 2048 <= LDS_BYTES && BN <= 256 && BNV <= NW, "pipelined epilogue must fit in the kernel's LDS");
    const int m_last = min(m0 + BM, M) - 1;
    const bool geglu = p.act == CS_ACT_GEGLU;
    const bool has_res = p.res != nullptr;
    // (r4: an edge column tile -- cout not a multiple of BN -- takes this path too, its lanes past cout masked like the rows
    // past M; only the fused gate needs whole [x | gate] column groups)
    // (r6: launches WITHOUT bias / residual / row vector -- the fused q | k | v projection, the Winograd position GEMMs -- take
    // this path too: its buffer stores with precomputed 32-bit offsets beat the plain path's per-store 64-bit address arithmetic,
    // 64.70 -> 64.41 ms per 32-object step same box, bit-identical, profiles/r06_h_piped_plain_ab.txt; -DCS_PIPED_PLAIN=0 = the r5
    // rule.  cs_conv_gemm_epilogue_caps keeps the narrower rule: no host asks such a launch for GroupNorm partials / a pair.)
#ifndef CS_PIPED_PLAIN
#define CS_PIPED_PLAIN 1
#endif
    const bool piped = !(CS_ABLATE & 32) && vec_epilogue && (p.res || p.bias || p.rowvec || CS_PIPED_PLAIN) && !p.scale && splits == 1 &&
                       (!geglu || n0 + BN <= p.cout) && (!p.rowvec || (m0 / p.rv_rows == m_last / p.rv_rows)) &&
                       (long long)ospan * p.ldo * 4 < 0x7FF00000LL && (!p.res || (long long)BM * p.ldr * 4 < 0x7FF00000LL);
    // r4 (ABI 14): what the epilogue emits beside / instead of the fp32 tile -- only this (piped) path can; the host asks
    // cs_conv_gemm_epilogue_caps first, so reaching another path with either set is a planning bug, reported loudly
    const bool gstat = p.gn_part != nullptr;                 // per-(tile, column) GroupNorm partial sums
    const bool opair = p.out_format == 2;                    // `out` as the interleaved operand pair of out * out_scale
    if ((gstat || opair) && !piped && tid == 0 && p.status) atomicOr(p.status, CS_STATUS_INTERNAL);
    if (piped) {
      __syncthreads();                                       // every wave has left the ring
      float* const ep = reinterpret_cast<float*>(smem + wave * EPB);
      // sum / sum of squares of the FINAL values of this lane's columns over the rows it stores: lane u = lane + 64 k keeps
      // float4 column (u % UPR) of staged row (u / UPR) in every pass, so 4 KU accumulator pairs cover the tile.  A lane
      // adds its 16 WMB values per column in fp32 (packed v_pk_add / v_pk_fma: a first version kept fp64 accumulators here
      // -- 768 double-rate VALU instructions per wave per tile, ~0.9 ms per step on an epilogue nothing overlaps); everything
      // above the lane -- rows of a wave, waves of a tile, tiles of a sample -- is summed in fp64.  Relative error of a
      // 16-term fp32 sum ~2e-7, averaged down by the thousands of lane sums in a group: fp64 GRADE while the group's |mean|
      // is of the order of its standard deviation (every GroupNorm input of the path: ratio <= ~3).  A group with |mean| >>
      // std loses (mean / std)^2 x 1e-7 of its variance to the cancellation E[x^2] - mean^2 on fp32-rounded lane sums (ratio
      // 30: rstd off by ~5e-5, tests/test_epilogue_outputs_gpu.py::test_partials_of_a_large_offset_tensor); the magnitude
      // bound derived from the same statistics stays an upper bound, and the overflow flag still backs it.
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      f32x2 gs[KU][2], gq[KU][2];
#pragma unroll
      for (int k = 0; k < KU; ++k)
#pragma unroll
        for (int e = 0; e < 2; ++e) gs[k][e] = gq[k][e] = f32x2{0.f, 0.f};
      float oamax = 0.f;                                     // largest |out * out_scale| this lane converted (opair)
      // interleaved pair: lanes 2t / 2t + 1 hold columns 8g .. 8g+3 / 8g+4 .. 8g+7 of one row (UPR is even); the even lane
      // stores [hi 8g .. 8g+7], the odd one [lo 8g .. 8g+7] -- one 16-byte store each after swapping a half (quad_perm 1,0,3,2)
      auto pair16 = [&](const f32x4& v) -> u32x4 {
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        h4 hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float o = v[e] * p.out_scale;
          oamax = fmaxf(oamax, fabsf(o));
          hi[e] = (_Float16)o;
          lo[e] = (_Float16)(o - (float)hi[e]);
        }
        const u32x2 H = __builtin_bit_cast(u32x2, hi), L = __builtin_bit_cast(u32x2, lo);
        const bool odd = lane & 1;
        const u32x2 send = odd ? H : L;
        u32x2 recv;
        recv[0] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)send[0], 0xB1, 0xF, 0xF, true);
        recv[1] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)send[1], 0xB1, 0xF, 0xF, true);
        u32x4 r;
        r[0] = odd ? recv[0] : H[0];
        r[1] = odd ? recv[1] : H[1];
        r[2] = odd ? L[0] : recv[0];
        r[3] = odd ? L[1] : recv[1];
        return r;
      };
      // byte offset of that store inside the row: 64-byte chunks of 16 columns = [hi 0-7 | lo 0-7 | hi 8-15 | lo 8-15]
      auto pair_off = [](int col) -> unsigned { return (unsigned)((col >> 4) * 64 + ((col & 8) ? 32 : 0) + ((col & 4) ? 16 : 0)); };
      float* const rs = reinterpret_cast<float*>(smem + RES0 + wave * 2 * RSLAB);
      float* const vb = reinterpret_cast<float*>(smem + VEC0);
      float* const vr = vb + 256;
      // per-tile descriptor windows (32-bit offsets inside BM rows)
      const long long o_skip = ob * p.ldo * 4, r_skip = (long long)m0 * p.ldr * 4;
      const long long o_left = ((Mo - 1) * p.ldo + p.cout) * 4 - o_skip;
      const long long r_left = has_res ? ((long long)(M - 1) * p.ldr + p.cout) * 4 - r_skip : 0;
      const long long o_cols = geglu ? p.cout / 2 : p.cout;
      const long long o_left2 = ((Mo - 1) * p.ldo + o_cols) * 4 - o_skip;
      (void)o_left;
      const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(
          (void*)((char*)p.out + o_skip), 0, o_left2 > 0x7FF00000LL ? 0x7FF00000u : (unsigned)o_left2, 0x00020000);
      const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(has_res ? (const char*)p.res + r_skip : (const char*)p.out), 0,
          r_left > 0x7FF00000LL ? 0x7FF00000u : (r_left > 0 ? (unsigned)r_left : 0u), 0x00020000);
      if (wave < BNV) {                                      // bias / row-vector rows: 4 bytes per lane
        const unsigned col = (unsigned)(64 * wave + lane);
        const unsigned off = col < (unsigned)BN ? (unsigned)(n0 + (int)col) * 4u : OOB;
        if (p.bias) {
          const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc((void*)p.bias, 0, (unsigned)p.cout * 4u, 0x00020000);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(brs, vb + 64 * wave, 4, off, 0, 0, 0);
        }
        if (p.rowvec) {
          const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc(
              (void*)(p.rowvec + (int64_t)(m0 / p.rv_rows) * p.ldrv), 0, (unsigned)p.cout * 4u, 0x00020000);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(vrs, vr + 64 * wave, 4, off, 0, 0, 0);
        }
      }
      auto pass_row = [&](int i, int q, int lrow) {
        return wm0 + 32 * i + 2 * (q & 1) + 8 * (q >> 1) + (lrow & 1) + 4 * (lrow >> 1);
      };
      auto fetch_res = [&](int pass) {                       // pass = i * NQ + q -> slab pass & 1
        const int i = pass / NQ, q = pass - i * NQ;
#pragma unroll
        for (int k = 0; k < KU; ++k) {
          const int u = lane + 64 * k;
          const int lrow = u / UPR, c4 = u - lrow * UPR;
          const int row = pass_row(i, q, lrow & 3);
          const unsigned off = (u < UN && m0 + row < M) ? (unsigned)(row * p.ldr + n0 + wn0 + 4 * c4) * 4u : OOB;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rrs, rs + (pass & 1) * (RSLAB / 4) + 256 * k, 16, off, 0, 0, 0);
        }
      };
      if (has_res) {
        fetch_res(0);
        wait_vmcnt<KU>();                                    // the bias / row-vector rows (older) have landed ...
      } else {
        wait_vmcnt<0>();
      }
      __syncthreads();                                       // ... for every wave
      constexpr int NPASS = WMB * NQ;
      auto do_pass = [&](auto i_c, auto q_c) {
        constexpr int i = decltype(i_c)::value, q = decltype(q_c)::value, pass = i * NQ + q;
        if (has_res) {
          if constexpr (pass + 1 < NPASS) fetch_res(pass + 1);
          // younger than this pass's residual fetch: the previous pass's KU stores and the KU fetches just issued
          wait_vmcnt<(pass > 0 ? KU : 0) + (pass + 1 < NPASS ? KU : 0)>();
        }
#pragma unroll
        for (int j = 0; j < WNB; ++j)
#pragma unroll
          for (int rr = 0; rr < QR; ++rr) ep[(rr + QR * half) * WCOLS + 32 * j + l31] = acc[i][j][QR * q + rr] * acc_scale_;
        if (geglu) {
          // columns of this wave = [x (WCOLS/2) | gate (WCOLS/2)] (weights packed that way by the host):
          // out[m][n/2 ..] = (x + bias_x) * gelu(gate + bias_g)   -- attention.py:44-46 fused into ff.net.0.proj
          constexpr int HC = WCOLS / 2, UPG = HC / 4 > 0 ? HC / 4 : 1, UNG = 2 * QR * UPG, KG = (UNG + 63) / 64;
#pragma unroll
          for (int k = 0; k < KG; ++k) {
            const int u = lane + 64 * k;
            const int lrow = u / UPG, c4 = u - lrow * UPG;
            const int row = pass_row(i, q, lrow & 3);
            f32x4 xv = {0.f, 0.f, 0.f, 0.f};
            unsigned off = OOB;
            const bool ok = u < UNG && m0 + row < M;
            if (ok) {
              xv = *reinterpret_cast<const f32x4*>(ep + lrow * WCOLS + 4 * c4);
              f32x4 gv = *reinterpret_cast<const f32x4*>(ep + lrow * WCOLS + HC + 4 * c4);
              if (p.bias) {
                xv += *reinterpret_cast<const f32x4*>(vb + wn0 + 4 * c4);
                gv += *reinterpret_cast<const f32x4*>(vb + wn0 + HC + 4 * c4);
              }
#pragma unroll
              for (int e = 0; e < 4; ++e) xv[e] = xv[e] * ((CS_ABLATE & 32768) ? gv[e] : cs_gelu(gv[e]));
              off = (unsigned)(orel(row) * p.ldo + (n0 + wn0) / 2 + 4 * c4) * 4u;
            }
            if (opair) {                                     // (uniform branch: every lane takes part in the half swap)
              const u32x4 pk = pair16(xv);
              if (ok) off = (unsigned)(orel(row) * p.ldo) * 4u + pair_off((n0 + wn0) / 2 + 4 * c4);
              if (CS_ABLATE & 16384) off = OOB;
              __builtin_amdgcn_raw_buffer_store_b128(pk, ors, off, 0, 0);
            } else {
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, xv), ors, off, 0, 0);
            }
          }
          return;
        }
#pragma unroll
        for (int k = 0; k < KU; ++k) {
          const int u = lane + 64 * k;
          const int lrow = u / UPR, c4 = u - lrow * UPR;
          const int row = pass_row(i, q, lrow & 3);
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
          unsigned off = OOB;
          const bool ok = u < UN && m0 + row < M && n0 + wn0 + 4 * c4 < p.cout;
          if (ok) {
            v = *reinterpret_cast<const f32x4*>(ep + lrow * WCOLS + 4 * c4);
            if (p.bias) v += *reinterpret_cast<const f32x4*>(vb + wn0 + 4 * c4);
            if (p.rowvec) v += *reinterpret_cast<const f32x4*>(vr + wn0 + 4 * c4);
            if (p.act != CS_ACT_NONE) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = cs_act(v[e], p.act);
            }
            if (has_res) v += *reinterpret_cast<const f32x4*>(rs + (pass & 1) * (RSLAB / 4) + 4 * u);
            off = (unsigned)(orel(row) * p.ldo + n0 + wn0 + 4 * c4) * 4u;
            if (gstat) {
              const f32x2 lo2 = {v[0], v[1]}, hi2 = {v[2], v[3]};
              gs[k][0] += lo2;
              gs[k][1] += hi2;
              gq[k][0] = __builtin_elementwise_fma(lo2, lo2, gq[k][0]);
              gq[k][1] = __builtin_elementwise_fma(hi2, hi2, gq[k][1]);
            }
          }
          if (CS_ABLATE & 16384) off = OOB;
          if (opair) {
            const u32x4 pk = pair16(v);
            if (ok && !(CS_ABLATE & 16384)) off = (unsigned)(orel(row) * p.ldo) * 4u + pair_off(n0 + wn0 + 4 * c4);
            __builtin_amdgcn_raw_buffer_store_b128(pk, ors, off, 0, 0);
          } else {
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ors, off, 0, 0);   // OOB lanes: dropped
          }
        }
      };
      auto all_q = [&](auto i_c) {
        do_pass(i_c, std::integral_constant<int, 0>{});
        do_pass(i_c, std::integral_constant<int, 1>{});
        do_pass(i_c, std::integral_constant<int, 2>{});
        do_pass(i_c, std::integral_constant<int, 3>{});
        do_pass(i_c, std::integral_constant<int, 4>{});
        do_pass(i_c, std::integral_constant<int, 5>{});
        do_pass(i_c, std::integral_constant<int, 6>{});
        do_pass(i_c, std::integral_constant<int, 7>{});
      };
      all_q(std::integral_constant<int, 0>{});
      if constexpr (WMB > 1) all_q(std::integral_constant<int, 1>{});
      static_assert(WMB <= 2 && NQ == 8, "pass enumeration");
      if (opair && p.status && oamax >= 65504.f) atomicOr(p.status, CS_STATUS_F16X3_OVERFLOW);
      if (gstat) {
        // lanes -> [wave][staged row][column] in LDS, then thread t < BN adds column t over the waves that own it and
        // the four staged rows, always in the same order; the tile's (sum, sum of squares) row goes out as 16-byte pairs
        static_assert(NW * 4 * WCOLS * 8 <= LDS_BYTES && BN <= NT, "statistics scratch must fit the kernel's LDS");
        // ONE round: every lane leaves its (sum, sum of squares) fp32 pairs, then 32 WAVES_M adds per column in fp64
        f32x2* const sc = reinterpret_cast<f32x2*>(smem);
        double tot[2] = {0.0, 0.0};
        __syncthreads();
#pragma unroll
        for (int k = 0; k < KU; ++k) {
          const int u = lane + 64 * k;
          if (u < UN) {
            const int lrow = u / UPR, c4 = u - lrow * UPR;
#pragma unroll
            for (int e = 0; e < 4; ++e)
              sc[(wave * 4 + lrow) * WCOLS + 4 * c4 + e] = f32x2{gs[k][e >> 1][e & 1], gq[k][e >> 1][e & 1]};
          }
        }
        __syncthreads();
        if (tid < BN) {
          const int wn = tid / WCOLS, cc = tid - wn * WCOLS;
          for (int wm = 0; wm < WAVES_M; ++wm)
#pragma unroll
            for (int lr = 0; lr < 4; ++lr) {
              const f32x2 v2 = sc[((wm * WAVES_N + wn) * 4 + lr) * WCOLS + cc];
              tot[0] += (double)v2[0];
              tot[1] += (double)v2[1];
            }
        }
        if (tid < BN && n0 + tid < p.cout) {
          typedef double d2 __attribute__((ext_vector_type(2)));
          const int tiles_m = (M + BM - 1) / BM;
          const int64_t trow = (int64_t)cls_id * tiles_m + tm;
          d2 o;
          o[0] = tot[0];
          o[1] = tot[1];
          *reinterpret_cast<d2*>(p.gn_part + (trow * p.gn_ld + n0 + tid) * 2) = o;
        }
      }
      return;
    }
  }
  const bool vec_ok = !(CS_ABLATE & 32) && vec_epilogue && (n0 + wn0 + WCOLS <= p.cout);
  if (vec_ok) {
    __syncthreads();                                        // every wave has left the ring
    float* ep = reinterpret_cast<float*>(smem + wave * EPI_BYTES);
    typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
    // (fused-reduce slices only: this tile's rows of the slice's partial tensor, 32-bit offsets)
    const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(outp + (int64_t)m0 * p.ldo), 0, (unsigned)((int64_t)BM * p.ldo * 4), 0x00020000);
#pragma unroll
    for (int i = 0; i < WMB; ++i)
#pragma unroll
      for (int ph = 0; ph < 16 / PASS_R; ++ph) {
#pragma unroll
        for (int j = 0; j < WNB; ++j)
#pragma unroll
          for (int rr = 0; rr < PASS_R; ++rr) {
            const int r = PASS_R * ph + rr;
            const int lrow = (r & 3) + 8 * ((r >> 2) % (PASS_R / 4)) + 4 * half;
            ep[lrow * WCOLS + 32 * j + l31] = acc[i][j][r] * acc_scale_;
          }
        // same-wave LDS ops are ordered; the compiler waits on lgkmcnt before the reads below
        if (p.act == CS_ACT_GEGLU) {
          // columns of this wave = [x (WCOLS/2) | gate (WCOLS/2)] (weights packed that way by the host):
          // out[m][n/2 ..] = (x + bias_x) * gelu(gate + bias_g)   -- attention.py:44-46 fused into ff.net.0.proj
          constexpr int HC = WCOLS / 2;
          constexpr int UNITS = EPI_ROWS * (HC / 4);
#pragma unroll
          for (int u0 = 0; u0 < UNITS; u0 += 64) {
            const int u = u0 + lane;
            if (u < UNITS) {
              const int lrow = u / (HC / 4);
              const int c4 = u - lrow * (HC / 4);
              const int m = m0 + wm0 + 32 * i + EPI_ROWS * ph + lrow;
              const int n = n0 + wn0 + 4 * c4;
              if (m < M) {
                f32x4 xv = *reinterpret_cast<const f32x4*>(ep + lrow * WCOLS + 4 * c4);
                f32x4 gv = *reinterpret_cast<const f32x4*>(ep + lrow * WCOLS + HC + 4 * c4);
                if (p.bias) {
                  xv += *reinterpret_cast<const f32x4*>(p.bias + n);
                  gv += *reinterpret_cast<const f32x4*>(p.bias + n + HC);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) xv[e] = xv[e] * cs_gelu(gv[e]);
                *reinterpret_cast<f32x4*>(outp + (ob + orel(m - m0)) * p.ldo + (n0 + wn0) / 2 + 4 * c4) = xv;
              }
            }
          }
          continue;
        }
        constexpr int UNITS = EPI_ROWS * (WCOLS / 4);
#pragma unroll
        for (int u0 = 0; u0 < UNITS; u0 += 64) {
          const int u = u0 + lane;
          if (u < UNITS) {
            const int lrow = u / (WCOLS / 4);
            const int c4 = u - lrow * (WCOLS / 4);
            const int m = m0 + wm0 + 32 * i + EPI_ROWS * ph + lrow;
            const int n = n0 + wn0 + 4 * c4;
            if (m < M) {
              f32x4 v = *reinterpret_cast<const f32x4*>(ep + lrow * WCOLS + 4 * c4);
              if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
              if (p.scale)
                v = v * *reinterpret_cast<const f32x4*>(p.scale + n) + *reinterpret_cast<const f32x4*>(p.shift + n);
              if (p.rowvec) v += *reinterpret_cast<const f32x4*>(p.rowvec + (int64_t)(m / p.rv_rows) * p.ldrv + n);
              if (p.act != CS_ACT_NONE) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = cs_act(v[e], p.act);
              }
              if (p.res) v += *reinterpret_cast<const f32x4*>(p.res + (int64_t)m * p.ldr + n);
              if (fz.sync)        // (uniform) a K slice of a fused-reduce launch: write-through, see fused_splitk_reduce
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4s, v), prs,
                                                       (unsigned)(((m - m0) * p.ldo + n) * 4), 0, 16);
              else
                *reinterpret_cast<f32x4*>(outp + (ob + orel(m - m0)) * p.ldo + n) = v;
            }
          }
        }
      }
    // r5: K slices of one output tile finish the reduce + epilogue here instead of in a second launch (see CsFuseK)
    if constexpr (BN <= NT && TPK == 9) {
      if (fz.sync) {
        __syncthreads();                                      // the staging LDS is free again
        fused_splitk_reduce<NT, BM, BN>(fz, p.out, smem, M, p.cout, m0, n0, splits, tm * tiles_n + tn, tid);
      }
    }
    return;
  }
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
          float v = acc[i][j][r] * acc_scale_ + bias;
          if (p.scale) v = v * sc + sh;
          if (p.rowvec) v += p.rowvec[(int64_t)(m / p.rv_rows) * p.ldrv + n];
          v = cs_act(v, p.act);
          if (p.res) v += p.res[(int64_t)m * p.ldr + n];
          outp[(ob + orel(row)) * p.ldo + n] = v;
        }
      }
    }
  }
}

template <int WMB, int WNB, int WAVES_M, int WAVES_N, bool PRE, int SLAB = 0, bool PAIR = false, int TPK = 9, bool PW = false>
int launch16(const CsConvGemm& p, int M, int splits, hipStream_t stream, int omap_f = 0, int omap_p = 0,
             const CsClsBatch* cls = nullptr, const CsFuseK* fuse = nullptr, int u_base = 0, int u_count = 0) {
  constexpr int BM = 32 * WMB * WAVES_M;
  constexpr int BN = 32 * WNB * WAVES_N;
  const int tiles_m = (M + BM - 1) / BM;
  const int tiles_n = (p.cout + BN - 1) / BN;
  // a unit range [u_base, u_base + u_count) of (position, column tile) pairs: the Winograd-W position launches only
  const int npos = (TPK == 3 && omap_p > 1) ? omap_p : 1;
  if (u_count < 0 || u_base < 0 || (u_count > 0 && (TPK != 3 || (cls && cls->n > 1) || fuse || tiles_m % npos ||
                                                    u_base + u_count > npos * tiles_n)))
    return CS_EINVAL;
  CsClsBatch cb;
  memset(&cb, 0, sizeof(cb));
  if (cls && TPK == 4 && cls->n > 1 && (omap_f & 7) && ((omap_f & 8) ? splits > 1 : splits == 1)) cb = *cls;
  const int64_t nblk = u_count > 0 ? (int64_t)u_count * (tiles_m / npos) * splits
                                   : (int64_t)tiles_m * tiles_n * splits * (cb.n > 1 ? cb.n : 1);
  if (nblk > 0x7fffffffLL) return CS_EINVAL;
  const int kg_per_tap = ((p.cin + 15) / 16) * 2;
  // buffer-descriptor extents: everything the loader may touch, and < 0xFFE00000 so OOB stays out of range
  const int64_t x_rows = (int64_t)p.nb * p.din * p.hin * p.win;
  const int64_t x_bytes = ((x_rows - 1) * p.lda + p.cin) * (PRE ? 2 : 4);
  const int64_t w_bytes = (int64_t)p.kd * p.kh * p.kw * kg_per_tap * p.cout * 16;
  if (w_bytes > 0xFFE00000LL) return CS_EINVAL;      // activations: per-workgroup windows, no 4 GiB limit
  // float4 epilogue needs 16-byte aligned rows in every operand it touches
  auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
  const int vec = (p.cout % 4 == 0) && (p.ldo % 4 == 0) && al16(p.out) && (!p.bias || al16(p.bias)) &&
                  (!p.scale || (al16(p.scale) && al16(p.shift))) &&
                  (!p.rowvec || (p.ldrv % 4 == 0 && al16(p.rowvec))) && (!p.res || (p.ldr % 4 == 0 && al16(p.res)));
  if (p.act == CS_ACT_GEGLU) {
    // fused gate: needs the float4 epilogue, whole [x | gate] column groups per wave and no other epilogue terms
    constexpr int WCOLS = 32 * WNB;
    if (!vec || (WCOLS / 2) % 4 || p.cout % WCOLS || p.scale || p.rowvec || p.res) return CS_EINVAL;
  }
  CsFuseK fz;
  memset(&fz, 0, sizeof(fz));
  if (fuse && fuse->sync) {
    // the slices' partial tiles must take the float4 staged store the fused tail hangs off, whole column tiles only
    if (splits < 2 || TPK != 9 || BN > 64 * WAVES_M * WAVES_N || !vec || p.cout % BN || p.bias || p.res || p.rowvec || p.scale ||
        p.act != CS_ACT_NONE || p.gn_part || p.out_format || fuse->reducers < 1 || fuse->reducers > splits ||
        BM % (16 * fuse->reducers))
      return CS_EINVAL;
    fz = *fuse;
  }
  CS_LAUNCH((conv_gemm_f16x3_kernel<WMB, WNB, WAVES_M, WAVES_N, PRE, SLAB, PAIR, TPK, PW>), dim3((unsigned)nblk), dim3(64 * WAVES_M * WAVES_N), 0,
            stream, p, M, tiles_n, p.kh * p.kw, p.kw, kg_per_tap, (long long)x_bytes, (unsigned)w_bytes, vec, splits,
            TPK == 4 ? omap_f : 0, TPK != 9 ? omap_p : 0, cb, fz, u_base, u_count);
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

// Will this descriptor run the four-taps-per-kd slab kernel (the 3x2x2 / 2x2x2 kernels of the Upsample convs folded onto
// the source grid)?  One rule for the dispatch below and for cs_conv_gemm_up2, which lets the classes store straight
// into the doubled grid only on that kernel.  CS_NO_SLAB4=1 keeps them on the per-tap gather (A/B runs, and the
// reference of the equality test).
bool cs_f16x3_slab4_ok(const CsConvGemm& p, int tile, int splits) {
#ifdef CS_NO_SLAB
  return false;
#else
  // (splits < 0: the geometry alone -- the K-sliced class batch of small launches, cs_conv_gemm_up2, r5)
  return !cs_debug()->no_slab4 && p.a_format == 0 && (tile == 4 || tile == 6) && splits <= 1 && p.kh == 2 && p.kw == 2 &&
         (p.kd == 2 || p.kd == 3) && p.sd == 1 && p.sh == 1 && p.sw == 1 && p.ud == 0 && p.uh == 0 && p.uw == 0 &&
         (unsigned)p.pd <= 1u && (unsigned)p.ph <= 1u && (unsigned)p.pw <= 1u && (p.kd == 2 || p.pd == 1) &&
         p.din == p.dout && p.hin == p.hout && p.win == p.wout && p.win <= 32 &&
         (256 + 2LL * p.hin * p.win + 2 * p.win + 32) * p.lda * 4 < 0x7FF00000LL;
#endif
}

// r5: is this the geometry of a Winograd-W position GEMM -- a 3x3x1 kernel over (D, H, W/2), stride 1, pads 1 / 1 / 0, same
// size out as in (cs_conv_gemm, a_format = 3, builds such a descriptor over the four stacked transformed operands)?
bool cs_f16x3_wino_geom(const CsConvGemm& p) {
  return p.kd == 3 && p.kh == 3 && p.kw == 1 && p.sd == 1 && p.sh == 1 && p.sw == 1 && p.pd == 1 && p.ph == 1 && p.pw == 0 &&
         p.ud == 0 && p.uh == 0 && p.uw == 0 && p.din == p.dout && p.hin == p.hout && p.win == p.wout && p.win <= 32 &&
         (512 + 2LL * p.hin * p.win + 2 * p.win + 32) * p.lda * 4 < 0x7FF00000LL;
}

// Line width of the A slab (32 / 64) the dispatch below stages for this descriptor on `tile` with `splits` K slices, 0 = the
// per-tap gather.  ONE rule: the dispatch asks it, and so does cs_conv_gemm_launch_info (what the hosts' per-kernel
// accounting reads instead of mirroring this file).
int cs_f16x3_slab_width(const CsConvGemm& p, int tile, int splits) {
#ifdef CS_NO_SLAB
  return 0;
#else
  if (splits < 1) splits = 1;
  if (p.a_format == 2) return 0;                                   // interleaved pairs: gather path only
  if (p.a_format == 0 && cs_f16x3_slab4_ok(p, tile, splits)) return 32;      // folded Upsample classes: four taps per kd
#ifndef CS_NO_SLAB
  if (cs_f16x3_wino_geom(p)) {                                               // Winograd-W position GEMMs: three taps per kd
    const int64_t nsc = 3LL * ((p.cin + 15) / 16);
    const bool ok = splits == 1 || ((nsc + splits - 1) / splits) * splits * 10 <= nsc * 11;
    return (p.a_format == 1 && (tile == 4 || tile == 6 || tile == 7) && ok) ? 32 : 0;
  }
#endif
  const bool geom0 = p.kd == 3 && p.kh == 3 && p.kw == 3 && p.sd == 1 && p.sh == 1 && p.sw == 1 && p.pd == 1 && p.ph == 1 &&
                     p.pw == 1 && p.ud == 0 && p.uh == 0 && p.uw == 0 && p.din == p.dout && p.hin == p.hout &&
                     p.win == p.wout && p.win <= 64 && (512 + 2LL * p.hin * p.win + 2 * p.win + 32) * p.lda * 4 < 0x7FF00000LL;
  // tiles 8 / 9 (512 x 64 / 512 x 128: two row blocks per wave) exist for the slab path on pre-split operands only (with the
  // in-loop fp32 -> hi/lo conversion two row blocks x nine unrolled taps spill); elsewhere they run as 7 / 6
  if (tile == 8 && !(geom0 && splits == 1 && p.a_format == 1)) tile = 7;
  if (tile == 9 && !(geom0 && splits == 1 && p.win <= 32 && p.a_format == 1)) tile = 6;
  const bool geom = (splits == 1 || tile == 4 || tile == 2 || tile == 6 || tile == 7) && geom0 &&
                    (p.win <= 32 || tile == 7 || tile == 8);
  // K slices of the slab kernel are whole super-chunks (nine taps): take it only where that granularity pads the slices
  // by at most a tenth (42 super-chunks over 32 slices would leave a third of the workgroups idle)
  const int64_t nsc_all = 3LL * ((p.cin + 15) / 16);
  const bool slices_ok = splits == 1 || ((nsc_all + splits - 1) / splits) * splits * 10 <= nsc_all * 11;
  if (!(geom && slices_ok)) return 0;
  switch (tile) {
    case 2: case 4: case 6: return 32;
    case 7: return p.win <= 32 ? 32 : 64;
    case 8: return p.a_format == 1 ? (p.win <= 32 ? 32 : 64) : 0;
    case 9: return p.a_format == 1 ? 32 : 0;
    default: return 0;
  }
#endif
}

// called from cs_conv_gemm (cs_gemm.hip) when desc->math == CS_MATH_F16X3; arguments already validated.
// omap_f / omap_p != 0: scattered store of one parity class of a folded Upsample conv (slab4 kernel only, see its epilogue)
// ncls > 1 (with omap_f): ONE launch covers all ncls parity classes -- class c has the packed weights cls_w[c] / cls_w_lo[c]
// and accumulator scale cls_acc[c]; its pads and scatter parity follow from c and omap_f (see the kernel)
int cs_conv_gemm_f16x3_dispatch(const CsConvGemm& p_in, int M, int tile, int splits, hipStream_t s, int omap_f, int omap_p,
                                const void* const* cls_w, const void* const* cls_w_lo, const float* cls_acc, int ncls,
                                const CsFuseK* fuse, int u_base, int u_count) {
  CsConvGemm p = p_in;
  if (u_count != 0 && !(p.a_format == 1 && cs_f16x3_wino_geom(p) && (omap_f & 16))) return CS_EINVAL;
  if (splits < 1) splits = 1;
  const bool cls_sliced = (omap_f & 8) != 0;           // r5: all parity classes x K slices in one launch, partial tiles out
  if (cls_sliced && (splits < 2 || ncls < 2 || p.a_format != 0 || (tile != 4 && tile != 6) || !cs_f16x3_slab4_ok(p, tile, -1) ||
                     p.bias || p.res || p.rowvec || p.scale || p.act != CS_ACT_NONE || p.gn_part || p.out_format))
    return CS_EINVAL;
  // (omap_f bit 4, r5: the launch covers omap_p Winograd-W position classes stacked along the batch -- no scatter)
  if ((omap_f & 15) && !cls_sliced && !cs_f16x3_slab4_ok(p, tile, splits)) return CS_EINVAL;
  CsClsBatch cb;
  memset(&cb, 0, sizeof(cb));
  if (ncls > 1) {
    if (!omap_f || ncls > 8 || !cls_w || !cls_w_lo || !cls_acc) return CS_EINVAL;
    for (int c = 0; c < ncls; ++c) {
      if (!cls_w[c] || !cls_w_lo[c] || !(cls_acc[c] > 0.f)) return CS_EINVAL;
      cb.w[c] = cls_w[c];
      cb.w_lo[c] = cls_w_lo[c];
      cb.acc_scale[c] = cls_acc[c];
    }
    cb.n = ncls;
  }
  if (p.a_scale == 0.f) p.a_scale = A_SCALE_DEFAULT;
  if (!p.w_lo || !(p.acc_scale > 0.f) || !(p.a_scale > 0.f)) return CS_EINVAL;
  if (p.kd * p.kh * p.kw > MAX_TAPS) return CS_EINVAL;                             // LDS row table extent
  if ((int64_t)p.nb * p.din * p.hin * p.win > 0x7fffffffLL) return CS_EINVAL;      // int32 row indices
  // 16-bit row deltas: the farthest tap is (kd-1) planes + (kh-1) rows + (kw-1) voxels from the window origin
  if ((int64_t)(p.kd - 1) * p.hin * p.win + (int64_t)(p.kh - 1) * p.win + p.kw > 32000) return CS_EINVAL;
  if (((uintptr_t)p.w & 15) || ((uintptr_t)p.w_lo & 15)) return CS_EINVAL;
  // tiles 8 / 9 (512 x 64 / 512 x 128: two row blocks per wave) exist for the slab path only; elsewhere they run as 7 / 6
  const bool slab_geom0 = p.kd == 3 && p.kh == 3 && p.kw == 3 && p.sd == 1 && p.sh == 1 &&
                          p.sw == 1 && p.pd == 1 && p.ph == 1 && p.pw == 1 && p.ud == 0 && p.uh == 0 && p.uw == 0 &&
                          p.din == p.dout && p.hin == p.hout && p.win == p.wout && p.win <= 64 &&
                          (512 + 2LL * p.hin * p.win + 2 * p.win + 32) * p.lda * 4 < 0x7FF00000LL;
  // (pre-split operands only: with the in-loop fp32 -> hi/lo conversion two row blocks x nine unrolled taps spill)
  if (tile == 8 && !(slab_geom0 && splits == 1 && p.a_format == 1)) tile = 7;
  if (tile == 9 && !(slab_geom0 && splits == 1 && p.win <= 32 && p.a_format == 1)) tile = 6;
  const bool slab_geom = cs_f16x3_slab_width(p, tile, splits) != 0 && !cs_f16x3_slab4_ok(p, tile, splits);   // (the one rule, above)
  const bool slab_slices_ok = true;
  if (p.a_format == 1) {
    if (!p.x_lo || ((uintptr_t)p.x_lo & 15) || (p.cin & 7) || (p.lda & 7)) return CS_EINVAL;
#ifndef CS_NO_SLAB
    if (cs_f16x3_wino_geom(p) && (tile == 4 || tile == 6 || tile == 7) && cs_f16x3_slab_width(p, tile, splits) == 32) {
      const int ncls = (omap_f & 16) ? omap_p : 0;
      if (ncls > 1 && (((M + 255) / 256) % ncls || M % 256)) return CS_EINVAL;     // whole row tiles per class
      // (256x224: the UNet's widths; 256x128 / 256x64: the VQ decoder's)
      if (tile == 4) return launch16<1, 7, 8, 1, true, 32, false, 3>(p, M, splits, s, 0, ncls, nullptr, nullptr, u_base, u_count);
      if (tile == 6) return launch16<1, 4, 8, 1, true, 32, false, 3>(p, M, splits, s, 0, ncls, nullptr, nullptr, u_base, u_count);
      return launch16<1, 2, 8, 1, true, 32, false, 3>(p, M, splits, s, 0, ncls, nullptr, nullptr, u_base, u_count);
    }
    if (slab_geom && slab_slices_ok) {
      switch (tile) {
        case 2: return launch16<1, 7, 4, 1, true, 32>(p, M, splits, s, 0, 0, nullptr, fuse);    // r3: small batches (128-row tiles, K slices)
        case 4: return launch16<1, 7, 8, 1, true, 32>(p, M, splits, s, 0, 0, nullptr, fuse);
        case 6: return launch16<1, 4, 8, 1, true, 32>(p, M, splits, s, 0, 0, nullptr, fuse);
        case 7: return p.win <= 32 ? launch16<1, 2, 8, 1, true, 32>(p, M, splits, s, 0, 0, nullptr, fuse)
                                   : launch16<1, 2, 8, 1, true, 64>(p, M, splits, s, 0, 0, nullptr, fuse);
        case 8: return p.win <= 32 ? launch16<2, 2, 8, 1, true, 32>(p, M, splits, s, 0, 0, nullptr, fuse)
                                   : launch16<2, 2, 8, 1, true, 64>(p, M, splits, s, 0, 0, nullptr, fuse);
        case 9: return launch16<2, 4, 8, 1, true, 32>(p, M, splits, s, 0, 0, nullptr, fuse);
        default: break;
      }
    }
#endif
    if (tile == 8) tile = 7;      // (only reachable in -DCS_NO_SLAB builds)
    if (tile == 9) tile = 6;
    switch (tile) {
      case 1: return launch16<2, 2, 2, 2, true>(p, M, splits, s, 0, 0, nullptr, fuse);
      case 2: return launch16<1, 7, 4, 1, true>(p, M, splits, s, 0, 0, nullptr, fuse);
      case 3: return launch16<1, 1, 2, 2, true>(p, M, splits, s, 0, 0, nullptr, fuse);
      case 4: return launch16<1, 7, 8, 1, true>(p, M, splits, s, 0, 0, nullptr, fuse);
      case 6: return launch16<1, 4, 8, 1, true>(p, M, splits, s, 0, 0, nullptr, fuse);
      case 7: return launch16<1, 2, 8, 1, true>(p, M, splits, s, 0, 0, nullptr, fuse);
      default: return CS_EINVAL;
    }
  }
  // pointwise (1x1x1, stride 1, no upsampling): no source-row tables (template argument PW); CS_NO_PW=1 for A/B runs
  const bool pw = !cs_debug()->no_pw && p.kd == 1 && p.kh == 1 && p.kw == 1 && p.sd == 1 && p.sh == 1 && p.sw == 1 &&
                  p.ud == 0 && p.uh == 0 && p.uw == 0 && p.pd == 0 && p.ph == 0 && p.pw == 0 &&
                  (int64_t)p.dout * p.hout * p.wout == (int64_t)p.din * p.hin * p.win &&
                  256LL * p.lda * 4 < 0x7FF00000LL;
  if (p.a_format == 2) {      // interleaved operand pair: same addressing as fp32, no conversion in the K loop
    if ((p.cin & 15) || (p.lda & 15)) return CS_EINVAL;
    if (tile == 8) tile = 7;
    if (tile == 9) tile = 6;
    if (pw) {
      switch (tile) {
        case 1: return launch16<2, 2, 2, 2, false, 0, true, 9, true>(p, M, splits, s, 0, 0, nullptr, fuse);
        case 2: return launch16<1, 7, 4, 1, false, 0, true, 9, true>(p, M, splits, s, 0, 0, nullptr, fuse);
        case 3: return launch16<1, 1, 2, 2, false, 0, true, 9, true>(p, M, splits, s, 0, 0, nullptr, fuse);
        case 4: return launch16<1, 7, 8, 1, false, 0, true, 9, true>(p, M, splits, s, 0, 0, nullptr, fuse);
        default: break;
      }
    }
    switch (tile) {
      case 1: return launch16<2, 2, 2, 2, false, 0, true>(p, M, splits, s, 0, 0, nullptr, fuse);
      case 2: return launch16<1, 7, 4, 1, false, 0, true>(p, M, splits, s, 0, 0, nullptr, fuse);
      case 3: return launch16<1, 1, 2, 2, false, 0, true>(p, M, splits, s, 0, 0, nullptr, fuse);
      case 4: return launch16<1, 7, 8, 1, false, 0, true>(p, M, splits, s, 0, 0, nullptr, fuse);
      case 6: return launch16<1, 4, 8, 1, false, 0, true>(p, M, splits, s, 0, 0, nullptr, fuse);
      case 7: return launch16<1, 2, 8, 1, false, 0, true>(p, M, splits, s, 0, 0, nullptr, fuse);
      default: return CS_EINVAL;
    }
  }
  if (p.a_format != 0) return CS_EINVAL;
  // r3: the 3x2x2 / 2x2x2 kernels of the Upsample convs folded onto the source grid (cs_conv_gemm_up2): slab path with
  // four taps per kd
  if (cs_f16x3_slab4_ok(p, tile, cls_sliced ? -1 : splits)) {
    if (tile == 4) return launch16<1, 7, 8, 1, false, 32, false, 4>(p, M, splits, s, omap_f, omap_p, &cb);
    return launch16<1, 4, 8, 1, false, 32, false, 4>(p, M, splits, s, omap_f, omap_p, &cb);
  }
#ifndef CS_NO_SLAB      // (A/B timing builds: -DCS_NO_SLAB keeps the per-tap gather everywhere)
  // 3x3x3, stride 1, "same" padding, no upsampling, one K slice, 256-row tiles: the A operand comes from a slab
  // shared by the nine (kh, kw) taps of each kd (see the kernel's header)
  const bool slab = slab_geom && slab_slices_ok;
  if (slab) {
    switch (tile) {
      case 2: return launch16<1, 7, 4, 1, false, 32>(p, M, splits, s, 0, 0, nullptr, fuse);   // small batches: 128-row tiles, K slices
      case 4: return launch16<1, 7, 8, 1, false, 32>(p, M, splits, s, 0, 0, nullptr, fuse);
      case 6: return launch16<1, 4, 8, 1, false, 32>(p, M, splits, s, 0, 0, nullptr, fuse);
      case 7: return p.win <= 32 ? launch16<1, 2, 8, 1, false, 32>(p, M, splits, s, 0, 0, nullptr, fuse)
                                 : launch16<1, 2, 8, 1, false, 64>(p, M, splits, s, 0, 0, nullptr, fuse);   // the decoder's 64^3 level
      default: break;
    }
  }
#endif
  if (pw) {
    switch (tile) {
      case 1: return launch16<2, 2, 2, 2, false, 0, false, 9, true>(p, M, splits, s, 0, 0, nullptr, fuse);
      case 2: return launch16<1, 7, 4, 1, false, 0, false, 9, true>(p, M, splits, s, 0, 0, nullptr, fuse);
      case 3: return launch16<1, 1, 2, 2, false, 0, false, 9, true>(p, M, splits, s, 0, 0, nullptr, fuse);
      case 4: return launch16<1, 7, 8, 1, false, 0, false, 9, true>(p, M, splits, s, 0, 0, nullptr, fuse);
      case 6: return launch16<1, 4, 8, 1, false, 0, false, 9, true>(p, M, splits, s, 0, 0, nullptr, fuse);
      case 7: return launch16<1, 2, 8, 1, false, 0, false, 9, true>(p, M, splits, s, 0, 0, nullptr, fuse);
      default: break;
    }
  }
  switch (tile) {
    case 1: return launch16<2, 2, 2, 2, false>(p, M, splits, s, 0, 0, nullptr, fuse);
    case 2: return launch16<1, 7, 4, 1, false>(p, M, splits, s, 0, 0, nullptr, fuse);
    case 3: return launch16<1, 1, 2, 2, false>(p, M, splits, s, 0, 0, nullptr, fuse);
    case 4: return launch16<1, 7, 8, 1, false>(p, M, splits, s, 0, 0, nullptr, fuse);
    case 6: return launch16<1, 4, 8, 1, false>(p, M, splits, s, 0, 0, nullptr, fuse);     // 256x128: the VQ decoder's 128- / 256-channel convs
    case 7: return launch16<1, 2, 8, 1, false>(p, M, splits, s, 0, 0, nullptr, fuse);     // 256x64:  its 64-channel convs
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
