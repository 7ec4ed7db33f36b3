/*
 * commonscenes_hip.h -- C ABI of libcommonscenes_hip.so (gfx950 / MI355X).
 *
 * This is the drop-in boundary for the CommonScenes shape-branch sampler
 * (SURVEY.md section 8b).  The reference has no native code on this path: every
 * entry below replaces a PyTorch/ATen call made from a reference Python file, cited
 * per entry as  <reference file>:<line>.  The only native precedent in the reference
 * (extension/chamfer_cuda.cpp:9-26, extension/chamfer.cu:136-151) returns an int
 * status and writes into caller-allocated outputs; this library keeps that contract.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer valid on the current HIP device;
 *   - the caller owns every buffer, including workspaces; the library allocates
 *     nothing and keeps no global state;
 *   - every entry enqueues on `stream` and never synchronises;
 *   - return value: 0 = success, CS_EINVAL = bad argument, otherwise hipError_t;
 *   - activations are channels-last: NDHWC, i.e. [n][d][h][w][c] with c contiguous and an
 *     explicit row stride `ld*` (in floats) so a tensor can live inside a wider
 *     (concatenated) buffer.  Token matrices [n][tokens][c] are the same memory.
 *   - all tensors are fp32 and every contraction accumulates in fp32.  The GEMM-shaped entries (cs_conv_gemm and its
 *     wrappers, cs_attn_selfattn_f16x3, the cs_unet_* / cs_vqvae_* drivers) have two numerics modes: CS_MATH_FP32
 *     (fp32 operands on v_mfma_f32_32x32x2_f32: bit-equal to an fp32 fma chain) and CS_MATH_F16X3 -- the mode the host
 *     classes and bench.py START in -- which carries every fp32 operand as an fp16 hi/lo pair and issues three
 *     v_mfma_f32_32x32x16_f16 per K = 16 step (~2^-22 per product, fp32-grade by measurement; operands must satisfy
 *     |a| * a_scale < 65504, reported through the CS_STATUS_F16X3_OVERFLOW word below).  cs_attn_selfattn_f16 (plain
 *     fp16 operands) is an opt-in reduced-precision mode.  GroupNorm statistics accumulate in fp64.
 */
#ifndef COMMONSCENES_HIP_H
#define COMMONSCENES_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* cs_stream_t; /* == hipStream_t */

#define CS_OK 0
#define CS_ENOMEM (-12)
#define CS_EINVAL (-22)

/* epilogue activation codes */
#define CS_ACT_NONE 0
#define CS_ACT_RELU 1
#define CS_ACT_SILU 2
#define CS_ACT_GELU 3 /* exact (erf) GELU, torch.nn.GELU() default */
#define CS_ACT_GEGLU 4 /* CS_MATH_F16X3 only: weight columns packed per 224-column tile as [x(112) | gate(112)];
                          out[m][n/2] = (x + bias) * gelu(gate + bias), out has cout/2 columns (attention.py:39-46) */

/* Sticky status bits a kernel may OR into a caller-owned device int32 (`status` arguments / CsConvGemm.status;
 * NULL = not wanted).  The library never clears the word and never reads it back.
 *   CS_STATUS_F16X3_OVERFLOW: a CS_MATH_F16X3 kernel met an operand with |a| * a_scale >= 65504 (fp16 range): its hi
 *   half is +-inf and the result of that launch is garbage -- re-run on CS_MATH_FP32 (the host classes do). */
#define CS_STATUS_F16X3_OVERFLOW 1
/*   CS_STATUS_INTERNAL: a kernel was asked for an epilogue output (gn_part / out_format) on a path that cannot produce it
 *   -- a host-side planning bug, never data dependent; the launch's extra outputs are missing. */
#define CS_STATUS_INTERNAL 2
/*   CS_STATUS_SPLITK_TIMEOUT (r6): a reducer of a fused split-K launch (CsConvGemm.splitk_sync) gave up waiting for a slice
 *   of its tile that never reached a CU -- the launch was not resident and the backstop fired (never data dependent).  That
 *   launch's output is incomplete and the counters may be left non-zero: zero splitk_sync and re-run with the two-kernel
 *   form (CsDebug.no_fused_reduce = 1 / splitk_sync = NULL) -- the host classes do. */
#define CS_STATUS_SPLITK_TIMEOUT 4

/*
 * Debug / A-B switches (r4: ONE struct instead of ~40 getenv() calls spread over two host languages).  All zero = the
 * product path -- the measured-best route everywhere.  cs_debug() parses the environment ONCE (field `x` <- CS_X in
 * capitals, e.g. CS_NO_SLAB4=1; "set, non-empty, not 0" for the flags) and every host reads this struct and nothing else:
 * the library itself, commonscenes_amd/ops.py (lib.debug()), csrc/cs_driver.h.  cs_debug_set overrides it at run time
 * (tests flip a switch inside one process; NULL re-reads the environment).  Not part of the data path's contract:
 * switches exist for same-box A/B timing and for the equality tests between two routes (every pair gives the same bits
 * unless a field says otherwise).
 */
typedef struct CsDebug {
  int32_t no_split16;         /* GroupNorm emits fp32, never the pre-split operand pair (a_format = 1) */
  int32_t no_pair16;          /* LayerNorm emits fp32, never the interleaved pair (a_format = 2) */
  int32_t no_upfold;          /* Upsample convs in direct form (27 taps on the doubled grid); different fp32 sum order */
  int32_t no_splitk;          /* hosts do not ask cs_conv_gemm_plan for K slices */
  int32_t no_fused_geglu;     /* GEGLU projection and gate as two kernels */
  int32_t no_tapcol;          /* thin-output convs on the implicit GEMM; different fp32 sum order */
  int32_t tapcol_tile;        /* tile of the taps-as-columns GEMM (0 = cs_tapcol_tile's rule) */
  int32_t no_cfg_split;       /* no channel-split ResBlocks; different fp32 sum order */
  int32_t concat_copy;        /* channel concatenation by copy (Python host) */
  int32_t tile512;            /* auto-select the 512-row slab tiles (they lost: DESIGN 4.5) */
  int32_t no_pw;              /* pointwise GEMMs on the generic gather kernel */
  int32_t no_slab4;           /* folded Upsample convs on the per-tap gather */
  int32_t no_attn_img;        /* attention without K / V tile images */
  int32_t attn_nw8;           /* eight-wave attention workgroups at every batch size */
  int32_t no_up2_direct;      /* folded Upsample classes through scratch + interleave */
  int32_t no_up2_batch;       /* one launch per Upsample parity class */
  int32_t plan_pow2;          /* power-of-two split-K plan (r2) */
  int32_t slice_tile2;        /* K-sliced slab convs on the 128-row tile */
  int32_t no_gn_parts;        /* GroupNorm statistics always from a pass over the tensor (r4) */
  int32_t no_pair_epilogue;   /* GEMM epilogues always write fp32 (r4) */
  int32_t no_dyn_scale;       /* raw-activation consumers keep the fixed operand scale 16 + overflow flag (r4) */
  int32_t no_tok_rules;       /* 1-tap GEMMs keep r3's tile / K-slice choices (r4: quantisation-aware tile, no slices under 128 chunks) */
  int32_t no_fused_reduce;    /* split-K always as two kernels (slices, then reduce + epilogue); same bits (r5) */
  int32_t no_gn_fold;         /* GroupNorm (mean, rstd) always by the separate finalize launch, never in the apply kernel's prologue (r5) */
  int32_t no_kwave;           /* small 1-tap GEMMs stay on the 64x64 one-accumulator-chain tile (r5: K cut across the four waves) */
  int32_t no_static_scales;   /* operands born inside a transformer block keep the constant scale 16 + overflow flag (r5: static bounds) */
  int32_t no_wino;            /* 3x3x3 convs always in direct form, never Winograd F(2,3) along W (r5, a_format = 3); different fp32 sums */
  int32_t wino_min_rows;      /* Winograd-W route from this many output rows (default 1024; 0 = the default) */
  int32_t no_wino43;          /* never F(4,3) along W (a_format = 4): F(2,3) wherever the Winograd-W route is taken */
  int32_t wino43_min_rows;    /* F(4,3) from this many output rows (default 2048; 0 = the default) */
  int32_t no_wino_tail;       /* Winograd-W position launches always with ONE slice count for every tile (r6: whole rounds unsliced +
                                 a K-sliced tail launch over the remaining (position, column tile) units); different fp32 sum order */
  int64_t split16_min_rows;   /* pre-split operands on the 128-row slab tile from this many rows (8192; 0 = never) */
  int64_t cfg_split_min_rows; /* channel-split ResBlocks from this many rows (65536) */
  int64_t gn_small_group;     /* single-launch GroupNorm up to this many elements per (sample, group) (11264) */
} CsDebug;
const CsDebug* cs_debug(void);
void cs_debug_set(const CsDebug* d);

/*
 * Host-side rules every host shares (csrc/cs_plan.hip; no device work).  r3 kept a copy of each in ops.py and in
 * cs_driver.h ("mirror of ops....") -- there is one now.
 *   cs_norm_a_scale        F16X3 operand scale of a GEMM fed by a GroupNorm / LayerNorm with max |gamma| = gmax, max |beta| =
 *                          bmax over n elements per statistic: the largest power of two that keeps |y| <= gmax sqrt(n - 1) +
 *                          bmax inside the fp16 range (cannot overflow whatever the input).
 *   cs_conv_wants_split16  should the GroupNorm feeding a (cout x cin x k^3) conv over m output rows emit the pre-split pair?
 *                          (plain = neither folded Upsample nor taps-as-columns)
 *   cs_tapcol_ok / cs_tapcol_tile   is a thin-output 3x3x3 conv run as taps-as-columns, and on which tile
 */
float cs_norm_a_scale(float gmax, float bmax, int64_t n);
/*   cs_bound_a_scale       (r5) the largest power of two s with bound * s <= 65000, clamped to [2^-24, 2^40]: the F16X3 operand
 *                          scale of a tensor whose magnitude is bounded by `bound` (static bounds of the operands born
 *                          inside a transformer block). */
float cs_bound_a_scale(float bound);
/*   cs_attnblock_static_scales  (ABI 18, r6) the F16X3 operand scales of an attention block whose q / k / v are
 *                          Conv1x1(GroupNorm(x)) + bias (vqvae_modules.py:154-178; openai_model_3d.py:360-366), from bounds
 *                          that hold for every input:  |GN(x)| <= E = gmax sqrt(n - 1) + bmax (n elements per group),
 *                          |q|, |k|, |v| <= B = w_l2max sqrt(c) E + b_absmax (w_l2max / b_absmax over ALL 3c rows of the fused
 *                          q | k | v weight / bias: cs_weight_rowstats), |attention output| <= B (softmax rows are convex
 *                          weights).  out4 = {scale of q * qk_scale, of k, of v, of the attention output -> proj_out}. */
int cs_attnblock_static_scales(float gmax, float bmax, int64_t n, int c, float w_l2max, float b_absmax, float qk_scale,
                               float* out4);
/*   cs_weight_rowstats     (r5; device work) out2 = {max over rows of ||row||_2 (rounded up), max |entry|} of a [rows][cols]
 *                          fp32 matrix, folded into out2 by atomicMax of the bits: zero it first.
 *   cs_transformer_static_scales  (r5) the F16X3 scales of the operands born inside a transformer block -- q / k / v of the
 *                          self-attention, its output (-> to_out), the GEGLU product (-> ff.net.2), t2 (-> proj_out) -- from
 *                          bounds that hold for EVERY input (attention.py:237-245, 335-351): see csrc/cs_plan.hip. */
typedef struct CsTransformerStats {
  float rq, rk, rv;       /* max row 2-norm of attn1.to_q / to_k / to_v */
  float ro, bo;           /* attn1.to_out.0: max row 2-norm, max |bias| */
  float rx, bx, rg, bg;   /* ff.net.0.proj value / gate halves: max row 2-norm, max |bias| */
  float r2, b2;           /* ff.net.2 */
  float rpi, bpi;         /* proj_in */
  float g1, be1, g3, be3; /* norm1 / norm3: max |gamma|, ||beta||_2 */
} CsTransformerStats;
int cs_weight_rowstats(const float* w, int rows, int cols, float* out2, cs_stream_t stream);
int cs_transformer_static_scales(const CsTransformerStats* st, int c, int64_t n_tokens, int heads, float gn_gmax,
                                 float gn_bmax, float ctx_max, float* out12);
int cs_conv_wants_split16(int64_t m, int cin, int cout, int k, int plain, int math);
int cs_tapcol_ok(int cout, int cin, int k, int math);
int cs_tapcol_tile(int64_t m, int ncolp);

/* GEMM numerics mode */
#define CS_MATH_FP32 0     /* v_mfma_f32_32x32x2_f32, bit-equal to an fp32 fma chain  */
#define CS_MATH_F16X3 1    /* 3x v_mfma_f32_32x32x16_f16 on hi/lo fp16 splits (~2^-21) */

/*
 * Implicit-GEMM convolution / linear descriptor.
 *   out[m][n] = act( (sum_{tap,c} x[src(m,tap)][c] * w[tap][c][n] + bias[n]) * scale[n] + shift[n]
 *                    + rowvec[m / rv_rows][n] ) + res[m][n]
 * with m = ((b*dout + od)*hout + oh)*wout + ow and
 *   src(m,tap): v = o*stride + k - pad in the "virtual" (nearest-upsampled) input of extent
 *   (din<<ud, hin<<uh, win<<uw); out-of-range taps read zero; physical coord = v >> u.
 * A plain Linear / 1x1x1 conv is kd=kh=kw=1, strides 1, pads 0, spatial extents 1.
 * Replaces: torch.nn.Conv3d / F.interpolate(nearest)+Conv3d / nn.Linear calls in
 *   model/networks/diffusion_networks/openai_model_3d.py:146-158,188-199,240-314,561,727
 *   model/networks/diffusion_networks/attention.py:42-46,163-170,313-328
 *   model/networks/vqvae_networks/vqvae_modules.py:24-39,64-123,128-152,337-399
 *   model/layers.py:21-38 (Linear + BatchNorm1d(eval) + ReLU via scale/shift/act)
 */
typedef struct CsConvGemm {
  const float* x;      /* [nb][din][hin][win][lda]                                  */
  const float* w;      /* [kd*kh*kw][cin][ldw]   (re-laid-out weights, n contiguous) */
  float* out;          /* [nb*dout*hout*wout][ldo]                                  */
  const float* bias;   /* [cout] or NULL */
  const float* scale;  /* [cout] or NULL (requires shift) */
  const float* shift;  /* [cout] or NULL */
  const float* rowvec; /* [ceil(M/rv_rows)][ldrv] or NULL */
  const float* res;    /* [M][ldr] or NULL */
  int32_t nb, din, hin, win;
  int32_t dout, hout, wout;
  int32_t cin, cout;
  int32_t lda, ldw, ldo, ldr, ldrv;
  int32_t kd, kh, kw;
  int32_t sd, sh, sw;
  int32_t pd, ph, pw;
  int32_t ud, uh, uw; /* log2 nearest-upsample factor applied to x before the conv */
  int32_t act;
  int32_t rv_rows;    /* rows per rowvec entry (tokens / voxels per sample) */
  int32_t math;       /* CS_MATH_* */
  int32_t tile;       /* 0 = auto, 1 = 128x128, 2 = 128x224, 3 = 64x64, 4 = 256x224 (F16X3; FP32 runs it as 2),
                         6 = 256x128, 7 = 256x64 (F16X3: channel counts that are not multiples of 224; FP32 runs them as 1 / 3),
                         8 = 512x64, 9 = 512x128 (F16X3, 3x3x3 stride-1 convs on pre-split operands: two row blocks per wave
                         over one A slab; any other call runs them as 7 / 6),
                         5 = (removed in r6) r2's persistent ping-pong kernel: a measured loser, never auto-selected; refused with
                         CS_EINVAL.  Every tile code gives the same bits. */
  /* CS_MATH_F16X3 only: w = hi halves, w_lo = lo halves, both laid out [tap][cin16/8][cout][8] by
   * cs_pack_weight_f16x3 (cin16 = cin rounded up to 16).  Activations are multiplied by a_scale (a power of
   * two; 0 means the default 16) before the fp16 split: |a| * a_scale must stay below 65504, and values
   * below 2^-14 / a_scale lose relative (not absolute) precision.  acc_scale = 1 / (weight_scale * a_scale). */
  const void* w_lo;
  float acc_scale;
  float a_scale;
  /* CS_MATH_F16X3, a_format = 1: the activations are already split -- x = fp16 hi image, x_lo = fp16 lo image,
   * both [rows][lda] halves holding value * a_scale (written by cs_groupnorm_apply_split16); cin, lda % 8 == 0.
   * a_format = 2 (ABI 12): x is the INTERLEAVED pair -- same bytes and lda (in floats) as the fp32 tensor, per row and
   * 16-channel chunk [hi c0-7 | lo c0-7 | hi c8-15 | lo c8-15] (cs_layernorm_pair16); cin, lda % 16 == 0; x_lo unused.
   * a_format = 3 (ABI 16): WINOGRAD-W form of a 3x3x3 stride-1 "same" conv (openai_model_3d.py:294-314: every ResBlock conv).
   *   F(2,3) along W only: per pair of output voxels (w = 2 w2, 2 w2 + 1) the four input voxels d0..d3 = w - 1 .. w + 2
   *   (zero outside the volume) become [d0 - d2, d1 + d2, d2 - d1, d1 - d3], the three kw taps g0..g2 of every (kd, kh)
   *   become [g0, (g0 + g1 + g2) / 2, (g0 - g1 + g2) / 2, g2], position q's products are summed over (kd, kh, cin) -- four
   *   GEMMs with a 3x3x1 kernel over (D, H, W/2), 18 instead of 27 multiply-adds per output -- and
   *   out[2 w2] = m0 + m1 + m2, out[2 w2 + 1] = m1 - m2 - m3.  Against fp64 the result is as close as the direct form's
   *   (2.5e-7 - 3.3e-7 rel-L2 at the UNet's shapes); it is NOT bit-equal to it.
   *   x / x_lo = the transformed operand, pre-split like a_format = 1: fp16 hi / lo images [4][nb][D][H][W/2][lda] of
   *   value * a_scale, written by cs_groupnorm_apply_wino16; w / w_lo = four consecutive packed images of the transformed
   *   weights (cs_pack_weight_f16x3_wino: ONE scale, so acc_scale as usual); din / hin / win, kd = kh = kw = 3 and every
   *   epilogue field describe the ORIGINAL conv; splitk_ws must hold cs_conv_wino_ws_bytes(desc) bytes (the four position
   *   results, x K slices if splitk > 1), the output transform + epilogue run in a second launch with the split-K reduce
   *   kernel's epilogue outputs (gn_part on 32-row tiles -- 64-row for a_format = 4 --, out_format).  cs_conv_wino_ok says where cs_conv_gemm accepts it.
   * a_format = 4 (ABI 17): the same with F(4,3) along W -- per FOUR output voxels the six inputs d0..d5 = w - 1 .. w + 4
   *   become B^T d (rows [4,0,-5,0,1,0], [0,-4,-4,1,1,0], [0,4,-4,-1,1,0], [0,-2,-1,2,1,0], [0,2,-1,-2,1,0], [0,4,0,-5,0,1]),
   *   the kw taps become G g (rows [1/4,0,0], [-1/6,-1/6,-1/6], [-1/6,1/6,-1/6], [1/24,1/12,1/6], [1/24,-1/12,1/6], [0,0,1]),
   *   SIX position GEMMs over (D, H, W/4) -- 13.5 of 27 multiply-adds per output -- and out[4 t + e] = (A^T m)_e with A^T rows
   *   [1,1,1,1,1,0], [0,1,-1,2,-2,0], [0,1,1,4,4,0], [0,1,-1,8,-8,1].  x / x_lo: images [6][nb][D][H][W/4][lda]
   *   (cs_groupnorm_apply_wino16_range with variant 4), w / w_lo: six packed images (cs_pack_weight_f16x3_wino variant 4).
   *   Against fp64 6e-7 - 8e-7 rel-L2 per conv at the UNet's shapes (direct form 3e-7): inside the per-op gate, ~2x the
   *   direct form's error -- which is why cs_conv_wino_ok grants it from a larger size only. */
  const void* x_lo;
  int32_t a_format;
  /* Split-K (CS_MATH_F16X3, 224-column tiles): splitk > 1 cuts the K loop (taps x channel chunks) into that many
   * slices, one workgroup each, so a GEMM with few output tiles (small batches: 12 tiles at one object's 256-voxel
   * level) still fills the 256 CUs.  Slices write fp32 partial tiles to splitk_ws ([splitk][M][cout] floats,
   * caller-owned, 16-byte aligned); a second kernel adds them in slice order -- deterministic -- and applies the
   * epilogue.  0 or 1 = off.  cs_conv_gemm_plan proposes the value. */
  int32_t splitk;
  void* splitk_ws;
  int32_t* status;   /* sticky CS_STATUS_* word (device), or NULL */
  /* ABI 14 (r4), CS_MATH_F16X3 only -- what the epilogue can emit BESIDE / INSTEAD OF the fp32 result, so that the next
   * operator does not have to read the tensor again for it (cs_conv_gemm_epilogue_caps says what a descriptor supports):
   *   gn_part != NULL: per (row tile, output column) the fp64 sum and sum of squares of the FINAL output values (after
   *     bias / row vector / activation / residual): gn_part[tile][gn_ld][2], tile = rows [tile * gn_rows, +gn_rows) of the
   *     output (folded Upsample classes: tile = class * tiles + source-row tile).  A GroupNorm that follows
   *     (ldm_diffusion_util.py:222-239, openai_model_3d.py:294-314) takes its statistics from these partials
   *     (cs_groupnorm_finalize_parts) instead of a pass over the tensor.  gn_rows must be the value
   *     cs_conv_gemm_epilogue_caps reports for this descriptor; gn_ld >= cout lets several producers fill column ranges of
   *     one buffer.  Sums run in a fixed order: bit-reproducible, independent of how many samples share the launch.
   *   out_format = 2: `out` receives the INTERLEAVED OPERAND PAIR of out * out_scale (the a_format = 2 layout above: same
   *     bytes and ldo as the fp32 tensor) -- for a result whose only reader is the next F16X3 GEMM (attention.py:241-244:
   *     GEGLU -> ff.net.2 -> proj_out).  |out| * out_scale >= 65504 raises CS_STATUS_F16X3_OVERFLOW.  0 = fp32. */
  double* gn_part;
  int32_t gn_ld;
  int32_t gn_rows;
  int32_t out_format;
  float out_scale;
  /*   a_bound != NULL (a_format = 0 only): the operand scale is taken from a BOUND on the tensor's magnitude instead of
   *     a_scale -- the kernel reads *a_bound (device float: max over the tensor's (sample, group) statistics of
   *     |mean| + std * sqrt(n - 1) >= max |x| by Samuelson's inequality, left by cs_groupnorm_finalize_parts /
   *     cs_groupnorm_parts, which have every group's mean and variance in hand anyway) and uses the largest power of two s
   *     with bound * s <= 65000 (clamped to [2^-8, 2^40]); acc_scale must be given for a_scale as usual, the kernel
   *     rescales it by the exact power of two.  For the consumers of RAW activations -- the residual stream read by
   *     skip_connection / Downsample / Upsample (openai_model_3d.py:146-199, 307-313) has no producer-side bound -- this
   *     replaces the fixed guess 16 and the CS_STATUS_F16X3_OVERFLOW detect-and-rerun cliff: no activation magnitude can
   *     leave the fp16 range, and a tensor of any scale keeps the same relative precision. */
  const float* a_bound;
  /* ABI 15 (r5), split-K only: splitk_sync != NULL lets cs_conv_gemm fold the reduce + epilogue INTO the slice kernel (no
   * second launch, no flush of the partial tiles between two kernels): every slice workgroup publishes its partial tile,
   * arrives on its output tile's counter, and the LAST `reducers` ARRIVERS of the tile (by the ticket of that atomic: r6 --
   * earlier arrivers leave at once and free their CUs) then each sum a 16-row-aligned share of
   * the tile over all slices IN SLICE ORDER and apply the epilogue -- the same sums, the same order, the same bits as the
   * two-kernel form (which remains the path when splitk_sync is NULL, when the launch has more workgroups than the device
   * can hold resident at once, or under CS_NO_FUSED_REDUCE=1).  splitk_sync points at splitk_sync_words int32 words
   * (>= 2 per output tile) that are ZERO on entry; the kernel returns them to zero, so one buffer zeroed once serves every
   * launch on a stream (launches on different streams need different buffers).  A slice that never arrives raises
   * CS_STATUS_SPLITK_TIMEOUT after a bounded wait instead of hanging (see that status bit). */
  int32_t* splitk_sync;
  int32_t splitk_sync_words;
} CsConvGemm;

int cs_conv_gemm(const CsConvGemm* desc, cs_stream_t stream);

/* What cs_conv_gemm's epilogue can emit for this descriptor (fill in everything that will be passed to cs_conv_gemm,
 * including splitk; gn_part / out_format themselves are ignored): *gn_rows = rows per statistics tile (0: not available
 * -- e.g. a tile would straddle two samples, or the launch takes the unfused epilogue), *pair_ok = 1 if out_format = 2 is
 * available.  Host-only.  ONE rule for every host (ops.py, cs_driver.h). */
int cs_conv_gemm_epilogue_caps(const CsConvGemm* desc, int32_t* gn_rows, int32_t* pair_ok);

/* Which kernel variant cs_conv_gemm will launch for this descriptor (fill in everything, including splitk): *tile = the
 * tile code (CsConvGemm.tile's numbering; 5 = removed), *slab = the line width of the A slab (32 / 64) or 0
 * for the per-tap gather.  Host-only; what bench.py's per-kernel accounting asks instead of mirroring the dispatch. */
int cs_conv_gemm_launch_info(const CsConvGemm* desc, int32_t* tile, int32_t* slab);

/* The split-K factor cs_conv_gemm's heuristic would pick for this descriptor (1 = none) and the workspace it then
 * needs; host-only, no device work.  A caller that wants it sets desc->splitk / desc->splitk_ws accordingly. */
int cs_conv_gemm_plan(const CsConvGemm* desc, int32_t* splitk, int64_t* splitk_ws_bytes);
/* (ABI 16) Winograd-W route (CsConvGemm.a_format = 3).  cs_conv_wino_ok: 1 if cs_conv_gemm takes this 3x3x3 conv in that
 * form -- desc as the conv would be issued in DIRECT form (a_format / x / w are not looked at): CS_MATH_F16X3, 3x3x3,
 * stride 1, pad 1, no upsampling, even W with W / 2 <= 32, cout % 224 == 0 (the UNet's widths: 256x224 tile) or cout % 128
 * == 0 / cout == 64 (the VQ decoder's: 256x128 / 256x64 tiles, never K-sliced -- its results must not depend on the batch),
 * cin % 8 == 0, whole 256-row tiles per position,
 * at least CsDebug.wino_min_rows output rows, CS_NO_WINO unset -- the ONE rule both hosts ask BEFORE they let the
 * GroupNorm emit the transformed operand.  cs_conv_wino_plan: the K slices of the position GEMMs (1 = none) and the bytes
 * of splitk_ws they need. */
/* (ABI 17: the return value is the VARIANT -- 0 = direct form, 2 = F(2,3) (a_format = 3), 4 = F(4,3) (a_format = 4: W % 4 == 0,
 * whole 256-row tiles per position over M / 4 rows, at least CsDebug.wino43_min_rows rows).  cs_conv_wino_plan
 * reads the variant from desc->a_format (3 or 4).) */
int cs_conv_wino_ok(const CsConvGemm* desc);
int cs_conv_wino_plan(const CsConvGemm* desc, int32_t* splitk, int64_t* ws_bytes);
/* (ABI 18) the shape of the position launch(es) behind that plan: `slices` = cs_conv_wino_plan's splitk; units_main > 0 = the
 * TAIL plan -- of the `units` (position, 224-column tile) units the first units_main run unsliced (whole rounds of the chip),
 * the rest as a second launch cut into `slices` K slices; units_main = 0 = every tile in `slices` uniform slices.  A unit's
 * rows are all summed the same way: results never depend on a sample's place in the batch.  Host-only (bench.py's byte
 * accounting); cs_conv_gemm takes the tail plan exactly when desc->splitk == slices. */
int cs_conv_wino_plan_info(const CsConvGemm* desc, int32_t* slices, int32_t* units_main, int32_t* units);
/* The two launches of cs_conv_gemm(a_format = 3) on their own, for hosts that time them separately (bench.py's per-kernel
 * HIP events): cs_conv_wino_positions = the four position GEMMs into splitk_ws, cs_conv_wino_output = the output transform +
 * epilogue from splitk_ws.  Same descriptor, same validation; calling the first and then the second IS cs_conv_gemm. */
int cs_conv_wino_positions(const CsConvGemm* desc, cs_stream_t stream);
int cs_conv_wino_output(const CsConvGemm* desc, cs_stream_t stream);

/*
 * Nearest x2 upsampling (in the dims flagged by ud / uh / uw, each 0 or 1) followed by a 3x3x3 stride-1 "same" conv
 * (Upsample: openai_model_3d.py:150-153 doubles H and W for dims=3, all three for dims=4; vqvae_modules.py:35-39 all
 * three), evaluated on the source grid: per output parity class the three taps of a doubled dim collapse to two source
 * taps with pre-summed weights, so the op is 4 / 8 convs with 3x2x2 / 2x2x2 kernels -- 12/27 resp. 8/27 of the direct
 * form's multiply-adds -- whose outputs interleave.
 *   cs_conv_up2_info         class count and folded kernel extents for a choice of doubled dims
 *   cs_fold_upsample_weight  w_torch [cout][cin][3][3][3] -> w_folded [cls][cout][cin][kd'][kh'][kw'] (fp32 sums in a
 *                            fixed order); class = (pd * nh + ph) * nw + pw.  Each class is then packed like any conv
 *                            weight (cs_relayout_weight / cs_pack_weight_f16x3 with taps = kd' * kh' * kw')
 *   cs_conv_gemm_up2         desc describes the WHOLE op exactly as cs_conv_gemm would take it (x on the source grid,
 *                            ud/uh/uw, dout = din << ud ..., out / ldo on the doubled grid, bias and activation
 *                            allowed; no residual / row vector / BN); desc->w, w_lo, acc_scale are ignored in favour
 *                            of the per-class arrays (host arrays of device pointers; w_lo_cls and acc_scale_cls only
 *                            for CS_MATH_F16X3).  ws: cs_conv_gemm_up2_ws_bytes(desc) bytes, 16-byte aligned.
 */
int cs_conv_up2_info(int ud, int uh, int uw, int32_t* ncls, int32_t* kd, int32_t* kh, int32_t* kw);
int cs_fold_upsample_weight(const float* w_torch, float* w_folded, int cout, int cin, int ud, int uh, int uw,
                            cs_stream_t stream);
int64_t cs_conv_gemm_up2_ws_bytes(const CsConvGemm* desc);
int cs_conv_gemm_up2(const CsConvGemm* desc, const void* const* w_cls, const void* const* w_lo_cls,
                     const float* acc_scale_cls, void* ws, cs_stream_t stream);

/* Convenience entries named in SURVEY.md section 8b (thin wrappers over cs_conv_gemm). */
int cs_conv3d_3x3x3_s111(const float* x, const float* w, const float* bias, float* out,
                         int nb, int d, int h, int w_, int cin, int cout, cs_stream_t stream);
int cs_conv3d_3x3x3_s122(const float* x, const float* w, const float* bias, float* out,
                         int nb, int d, int h, int w_, int cin, int cout, cs_stream_t stream);
int cs_gemm_tokens(const float* x, const float* w, const float* bias, const float* res, float* out,
                   int m, int k, int n, int act, cs_stream_t stream);

/*
 * Weight re-layout: torch (cout, cin, kd, kh, kw) -> [tap][cin_pad][ldw] (zero padded).
 * Also used for Linear (out,in) with taps = 1.
 */
int cs_relayout_weight(const float* w_torch, float* w_out, int cout, int cin, int taps,
                       int cin_pad, int ldw, cs_stream_t stream);

/*
 * CS_MATH_F16X3 weight packing.  Every fp32 value v is carried as two fp16 halves of v' = v * 2^s:
 *   hi = fp16(v'), lo = fp16(v' - hi)  (22 mantissa bits together), and the contraction evaluates
 *   a.w ~= (a_hi.w_hi + a_hi.w_lo + a_lo.w_hi) * 2^-(s_a + s_w) with three v_mfma_f32_32x32x16_f16 per
 * K=16 step, fp32 accumulation (each fp16 x fp16 product is exact in fp32).  Activations are split on the
 * fly inside the GEMM (scale CsConvGemm.a_scale, default 2^4); `scale` = 2^s_w is chosen by the caller so that
 * max|w| * scale <= 2^14.  Output layout: [tap][cin16/8][cout][8] halves, cin16 = round_up(cin, 16).
 */
int cs_pack_weight_f16x3(const float* w_torch, void* w_hi, void* w_lo, int cout, int cin, int taps,
                         float scale, cs_stream_t stream);
/* (ABI 16) The Winograd-W weights of a 3x3x3 conv (CsConvGemm.a_format = 3): w_torch [cout][cin][3][3][3] -> FOUR packed
 * images (position q = 0..3, each [9 taps (kd, kh)][cin16/8][cout][8] halves, q-major) of
 * u_q = [g0, (g0 + g1 + g2) / 2, (g0 - g1 + g2) / 2, g2] over the kw taps g0..g2 (formed in fp64, rounded once) * scale;
 * max |u_q| <= 1.5 max |w|: choose scale with that headroom. */
int cs_pack_weight_f16x3_wino(const float* w_torch, void* w_hi, void* w_lo, int cout, int cin, float scale,
                              cs_stream_t stream);
/* (ABI 17) variant = 2: the entry above; variant = 4: SIX images of G g (F(4,3), max |u_q| <= max |w|); src_cin / c0: the
 * packed weight is input channels [c0, c0 + cin) of a [cout][src_cin][27] tensor (0 / 0 = the whole tensor). */
int cs_pack_weight_f16x3_wino_v(const float* w_torch, void* w_hi, void* w_lo, int cout, int cin, float scale, int variant,
                                int src_cin, int c0, cs_stream_t stream);

/*
 * Thin-output 3x3x3 convs (cout <= 4, stride 1, "same" padding: openai_model_3d.py:733-737 `self.out`,
 * vqvae_modules.py:473 `conv_out`) as "taps as columns" (r3): one POINTWISE GEMM with 27 * cout columns,
 *   Y[m'][o * 27 + t] = sum_c A[m'][c] * W[o][c][t],  then  out[m][o] = bias[o] + sum_t Y[m + off_t][o * 27 + t]
 * with taps that leave the volume contributing 0.  The implicit GEMM would spend a 64-column tile on 1-4 real columns.
 *   cs_pack_weight_f16x3_tapcol: torch (cout, cin, 3, 3, 3) weight -> the cs_pack_weight_f16x3 layout (taps = 1) of
 *     the [ncolp][cin] pointwise weight, row o * 27 + t = W[o][:][t]; ncolp >= 27 * cout, a multiple of 4, extra rows 0.
 *   cs_tapsum27: y [nb*d*h*w][ldy] (the GEMM's output, no bias) -> out [nb*d*h*w][ldo]; bias [cout] or NULL.  Sum in
 *     tap order t = 0 .. 26 (kd, kh, kw row-major), then the bias: bit-reproducible.
 */
int cs_pack_weight_f16x3_tapcol(const float* w_torch, void* w_hi, void* w_lo, int cout, int cin, int ncolp, float scale,
                                cs_stream_t stream);
int cs_tapsum27(const float* y, const float* bias, float* out, int nb, int d, int h, int w, int cout, int ldy, int ldo,
                cs_stream_t stream);

/*
 * GroupNorm over NDHWC. Two entries: statistics (fp64 accumulation), then normalise+affine+act.
 *   ws: workspace of cs_groupnorm_ws_bytes(nb, groups) bytes.
 *   stats: [nb][groups][2] floats (mean, rstd).
 * Replaces GroupNorm32 (ldm_diffusion_util.py:222-239), Normalize (attention.py:78-79,
 * vqvae_modules.py:13-21) followed by SiLU / swish / GELU / identity.
 */
int64_t cs_groupnorm_ws_bytes(int nb, int groups);
int cs_groupnorm_stats(const float* x, int nb, int rows, int c, int ldx, int groups, float eps,
                       void* ws, float* stats, cs_stream_t stream);
/* (ABI 14) ... and the tensor's magnitude bound beside the statistics: see cs_groupnorm_finalize_parts / CsConvGemm.a_bound */
int cs_groupnorm_stats_bound(const float* x, int nb, int rows, int c, int ldx, int groups, float eps, void* ws,
                             float* stats, float* bound, cs_stream_t stream);
int cs_groupnorm_apply(const float* x, const float* stats, const float* gamma, const float* beta,
                       float* y, int nb, int rows, int c, int ldx, int ldy, int groups, int act,
                       cs_stream_t stream);
/* Same as cs_groupnorm_apply but the result is emitted as the fp16 hi / lo pair of y * a_scale (two [rows][ldy]
 * half images), the A-operand format of CS_MATH_F16X3 GEMMs with a_format = 1. */
int cs_groupnorm_apply_split16(const float* x, const float* stats, const float* gamma, const float* beta,
                               void* y_hi, void* y_lo, int nb, int rows, int c, int ldx, int ldy, int groups,
                               int act, float a_scale, int32_t* status, cs_stream_t stream);
/* (ABI 16) ... emitted in the Winograd-W form (CsConvGemm.a_format = 3): the four transformed operands [d0 - d2, d1 + d2,
 * d2 - d1, d1 - d3] of y = act(GroupNorm(x)) along W (y = 0 outside the volume), each split into fp16 hi / lo of value *
 * a_scale -- v_hi / v_lo: [4][nb][d][h][w / 2][ldv] halves.  |value| <= 2 max|y|: a_scale must leave that headroom
 * (cs_norm_a_scale(...) / 2).  w even, c % 8 == 0, ldv % 8 == 0. */
int cs_groupnorm_apply_wino16(const float* x, const float* stats, const float* gamma, const float* beta, void* v_hi,
                              void* v_lo, int nb, int d, int h, int w, int c, int ldx, int ldv, int groups, int act,
                              float a_scale, int32_t* status, cs_stream_t stream);
/* ... and its channel-range form (see cs_groupnorm_apply_split16_range): channels ch0 .. ch0 + c of a tensor whose statistics
 * were taken over `groups` groups of `cpg` channels; x, gamma, beta point AT channel ch0 (the channel-split ResBlocks). */
int cs_groupnorm_apply_wino16_range(const float* x, const float* stats, const float* gamma, const float* beta, void* v_hi,
                                    void* v_lo, int nb, int d, int h, int w, int c, int ldx, int ldv, int groups, int cpg,
                                    int ch0, int act, float a_scale, int32_t* status, cs_stream_t stream);
/* (ABI 17) ... with the transform as an argument: variant = 2 is the entry above (F(2,3): four images over W / 2), variant = 4
 * emits the SIX F(4,3) images [6][nb][d][h][w / 4][ldv] (w % 4 == 0; |value| <= 10 max|y|: a_scale = cs_norm_a_scale(...) / 16). */
int cs_groupnorm_apply_wino_range(const float* x, const float* stats, const float* gamma, const float* beta, void* v_hi,
                                  void* v_lo, int nb, int d, int h, int w, int c, int ldx, int ldv, int groups, int cpg,
                                  int ch0, int act, float a_scale, int variant, int32_t* status, cs_stream_t stream);
/* Channel-range forms of the two apply entries (ABI 11): the c channels handled are channels ch0 .. ch0 + c of a tensor
 * whose statistics were taken over `groups` groups of `cpg` channels (stats: [nb][groups][2]); x, gamma, beta and the
 * outputs point AT channel ch0 (ch0 % 4 == 0).  One statistics pass over a channel concatenation [h | skip]
 * (openai_model_3d.py:781 followed by in_layers, :294-300) can then feed separate GEMM operands per channel range:
 * under classifier-free guidance the skip half of output blocks 5-8 is the same tensor for both guidance halves
 * (samplers/ddim.py:206-209 duplicates x), so its part of the convolution is evaluated once (unet.py::_res_split). */
int cs_groupnorm_apply_range(const float* x, const float* stats, const float* gamma, const float* beta, float* y,
                             int nb, int rows, int c, int ldx, int ldy, int groups, int cpg, int ch0, int act,
                             cs_stream_t stream);
int cs_groupnorm_apply_split16_range(const float* x, const float* stats, const float* gamma, const float* beta,
                                     void* y_hi, void* y_lo, int nb, int rows, int c, int ldx, int ldy, int groups,
                                     int cpg, int ch0, int act, float a_scale, int32_t* status, cs_stream_t stream);
/* Statistics + normalisation + activation in one call (what the hosts use): a single launch (one workgroup per
 * (sample, group), fp64 sums in a fixed order) while the tensor is at most 16 MB and a group at most 11264 elements
 * (one or two objects at the 16x4x4 level: there the three launches above sit at their launch floors), otherwise cs_groupnorm_stats +
 * cs_groupnorm_apply.  `stats` is written either way; `ws` as for cs_groupnorm_stats. */
int cs_groupnorm(const float* x, const float* gamma, const float* beta, float* y, int nb, int rows, int c, int ldx,
                 int ldy, int groups, float eps, int act, void* ws, float* stats, cs_stream_t stream);
/* (ABI 14, r4) GroupNorm statistics WITHOUT a pass over the tensor: from the per-(row tile, column) fp64 partial sums the
 * producing GEMMs' epilogues wrote (CsConvGemm.gn_part).  The c channels of the normalised tensor are covered by 1-4
 * column segments, one per producer -- a channel concatenation [h | skip] (openai_model_3d.py:781) has two; a skip tensor
 * that one copy serves for both classifier-free-guidance halves is described by nb_src < nb (sample n reads the
 * partials of sample n % nb_src).  One wave per (sample, group) adds the group's partials in a fixed order (lanes
 * stride over the tiles of one channel after the other, then a butterfly): bit-reproducible, fp64 throughout, the same
 * mean / rstd expressions as cs_groupnorm_stats.  `stats`: [nb][groups][2] as for cs_groupnorm_apply* (NULL: not wanted).
 * `bound` (NULL or a device float the caller zeroed): receives max over (sample, group) of |mean| + std * sqrt(n - 1), an
 * upper bound of max |x| over the whole tensor (Samuelson) -- what CsConvGemm.a_bound reads; an atomic max of the value's
 * bits, order-independent. */
typedef struct CsGnSeg {
  const double* part;        /* [tiles][ld][2] (sum, sum of squares) */
  int32_t ld;                /* columns per tile of `part` */
  int32_t col0;              /* column of `part` holding the segment's first channel */
  int32_t ch0;               /* first channel of the normalised tensor this segment covers */
  int32_t nch;               /* channels covered */
  int32_t tiles_per_sample;  /* statistics tiles per sample (and per class): rows per sample / CsConvGemm.gn_rows */
  int32_t ncls;              /* 1; or the parity-class count of a folded Upsample launch (cs_conv_gemm_up2), whose tiles are
                                ordered [class][sample][tile] over the SOURCE rows */
  int32_t nb_src;            /* samples the producer ran */
  int32_t reserved;
} CsGnSeg;
int cs_groupnorm_finalize_parts(const CsGnSeg* segs, int nseg, int nb, int rows, int c, int groups, float eps,
                                float* stats, float* bound, cs_stream_t stream);
/* ... and the whole GroupNorm (+ activation) from them in one call, the counterpart of cs_groupnorm: ONE launch for small
 * tensors (the same size rule: one workgroup per (sample, group) adds the group's partials and makes a single sweep over
 * it), cs_groupnorm_finalize_parts + cs_groupnorm_apply otherwise.  `stats` is written either way. */
int cs_groupnorm_parts(const float* x, const CsGnSeg* segs, int nseg, const float* gamma, const float* beta, float* y,
                       int nb, int rows, int c, int ldx, int ldy, int groups, float eps, int act, float* stats,
                       float* bound, cs_stream_t stream);
/* SURVEY name: cs_groupnorm with a SiLU epilogue. */
int cs_groupnorm_silu_ndhwc(const float* x, const float* gamma, const float* beta, float* y,
                            int nb, int rows, int c, int groups, float eps, void* ws, float* stats,
                            cs_stream_t stream);

/* LayerNorm over the last dim of [m][c] (nn.LayerNorm, attention.py:229-231). */
int cs_layernorm(const float* x, const float* gamma, const float* beta, float* y, int m, int c,
                 int ldx, int ldy, float eps, cs_stream_t stream);

/* LayerNorm whose output is the INTERLEAVED F16X3 operand pair of y * a_scale (ABI 12): y has the bytes and row stride of
 * the fp32 [m][ldy] tensor it replaces -- per row and 16-channel chunk the 64 bytes [hi c0-7 | lo c0-7 | hi c8-15 |
 * lo c8-15] (fp16 halves) -- and feeds cs_conv_gemm with CsConvGemm.a_format = 2 (x = y, same lda, a_scale as here):
 * the consuming GEMM (the fused q|k|v and GEGLU projections behind attention.py:229-231,237-245) then gathers the same
 * 64-byte pieces as from an fp32 tensor and its K loop carries no fp32 -> hi/lo conversion.  c, ldy % 16 == 0. */
int cs_layernorm_pair16(const float* x, const float* gamma, const float* beta, void* y, int m, int c, int ldx, int ldy,
                        float eps, float a_scale, int32_t* status, cs_stream_t stream);

/*
 * Multi-head attention, flash style (no score matrix in HBM), fp32.
 *   q: [nb][nq][ldq], head h at columns [h*dh, (h+1)*dh);  k, v likewise over nk keys.
 *   out[b][i][h*dh + d] = sum_j softmax_j(scale * q_i . k_j) v_j[d]
 * Replaces CrossAttention.forward einsum/softmax/einsum (attention.py:201-218) and
 * AttnBlock.forward bmm/softmax/bmm (vqvae_modules.py:162-173).
 */
int cs_attn_selfattn(const float* q, const float* k, const float* v, float* out, int nb, int nq,
                     int nk, int heads, int dh, int ldq, int ldk, int ldv, int ldo, float scale,
                     cs_stream_t stream);

/* Same contract, CS_MATH_F16X3 numerics: Q, K, V and the probabilities are carried as fp16 hi/lo pairs on the
 * fp16 MFMA (3 products per contraction, fp32 accumulate and fp32 softmax). */
int cs_attn_selfattn_f16x3(const float* q, const float* k, const float* v, float* out, int nb, int nq,
                           int nk, int heads, int dh, int ldq, int ldk, int ldv, int ldo, float scale,
                           int32_t* status, cs_stream_t stream);

/* (ABI 15, r5) cs_attn_selfattn_f16x3 with caller-chosen power-of-two operand pre-scales for Q * scale, K and V (the plain
 * entry uses 16 for all three): see csrc/cs_attention_f16x3.hip. */
int cs_attn_selfattn_f16x3_scaled(const float* q, const float* k, const float* v, float* out, int nb, int nq, int nk,
                                  int heads, int dh, int ldq, int ldk, int ldv, int ldo, float scale, float q_scale,
                                  float k_scale, float v_scale, int32_t* status, cs_stream_t stream);
/* Workspace form (r3, ABI 13): with `ws` (cs_attn_f16x3_ws_bytes(...) bytes, 16-byte aligned) K and V are split into
 * their fp16 hi / lo tile images ONCE per call by a pre-pass and the attention kernel streams whole images into LDS with
 * buffer_load ... lds -- the plain entry converts every K / V tile inside every workgroup.  Covers 128 < dh <= 256 from
 * 1024 queries up (the VQ decoder's single 256-channel head over 4096 tokens: 2.3x); cs_attn_f16x3_ws_bytes returns 0
 * for every other shape, and ws == NULL or such a shape runs the plain kernel.  Bit-identical results either way. */
int64_t cs_attn_f16x3_ws_bytes(int nb, int nq, int nk, int heads, int dh);
int cs_attn_selfattn_f16x3_ws(const float* q, const float* k, const float* v, float* out, int nb, int nq,
                              int nk, int heads, int dh, int ldq, int ldk, int ldv, int ldo, float scale,
                              int32_t* status, void* ws, cs_stream_t stream);
/* (ABI 18, r6) the workspace form with caller-chosen operand pre-scales (cs_attn_selfattn_f16x3_scaled's arguments): what an
 * attention block fed by a GroupNorm takes -- the VQ decoder's AttnBlock (vqvae_modules.py:154-178), the concat family's
 * AttentionBlock (openai_model_3d.py:360-366) -- with the scales of cs_attnblock_static_scales. */
int cs_attn_selfattn_f16x3_ws_scaled(const float* q, const float* k, const float* v, float* out, int nb, int nq,
                                     int nk, int heads, int dh, int ldq, int ldk, int ldv, int ldo, float scale,
                                     float q_scale, float k_scale, float v_scale, int32_t* status, void* ws,
                                     cs_stream_t stream);

/* Same contract, PLAIN fp16 operands on the fp16 MFMA (one pass instead of three; fp32 softmax and accumulation):
 * the "fp16 MFMA attention" option BASELINE configs[4] names.  Reduced precision (~3e-4 relative on the attention
 * output) -- opt-in, outside the fp32 parity gates, never the default. */
int cs_attn_selfattn_f16(const float* q, const float* k, const float* v, float* out, int nb, int nq,
                         int nk, int heads, int dh, int ldq, int ldk, int ldv, int ldo, float scale,
                         int32_t* status, cs_stream_t stream);

/* GEGLU gate: out[m][j] = x[m][j] * gelu(x[m][h + j])  (attention.py:44-46). */
int cs_geglu(const float* x, float* out, int m, int h, int ldx, int ldo, cs_stream_t stream);

/* dst[m][0:c] = src[m][0:c] with independent row strides (channel concat, openai_model_3d.py:781). */
int cs_copy_rows(const float* src, float* dst, int64_t m, int c, int lds, int ldd,
                 cs_stream_t stream);

/* x[m][c] += v[m / rows][c]   (cross-attention with one context token). */
int cs_add_rowvec(float* x, const float* v, int64_t m, int c, int ldx, int ldv, int rows,
                  cs_stream_t stream);

/* Layout: NCDHW [nb][c][s] -> NDHWC [nb][s][cpad] (channels >= c zero filled) and back. */
int cs_nchw_to_ndhwc(const float* x, float* y, int nb, int c, int s, int cpad, cs_stream_t stream);
/* (ABI 18) max |x| over n floats folded into *slot (atomicMax of the bits: zero it first).  What both hosts leave as the
 * magnitude bound (CsConvGemm.a_bound) of the UNet's conv_in operand -- the RAW latent x_t, which no normalisation bounds
 * (openai_model_3d.py:752-766, input_blocks[0]): its F16X3 operand scale then follows the tensor, no overflow possible. */
int cs_absmax(const float* x, int64_t n, float* slot, cs_stream_t stream);
int cs_ndhwc_to_nchw(const float* x, float* y, int nb, int c, int s, int ldx, cs_stream_t stream);

/*
 * Sinusoidal timestep embedding (ldm_diffusion_util.py:174-194):
 *   out[b][0:half] = cos(t[b]*f_k), out[b][half:2*half] = sin(t[b]*f_k), f_k = exp(-ln(max_period)*k/half).
 */
int cs_timestep_embedding(const int64_t* t, float* out, int nb, int dim, float max_period,
                          cs_stream_t stream);

/*
 * Fused classifier-free guidance + DDIM update (samplers/ddim.py:206-243), eta == 0 path plus
 * optional noise term.  eps: [2*nb][per] with the unconditional half first.
 *   e      = e_uc + scale * (e_c - e_uc)
 *   pred   = (x - sqrt_one_minus_at * e) / sqrt(a_t)
 *   x_prev = sqrt(a_prev) * pred + sqrt(1 - a_prev - sigma^2) * e + sigma * noise
 * x_prev may alias x.  pred_x0 and noise may be NULL.  If cfg == 0, eps is [nb][per].
 */
int cs_ddim_cfg_update(const float* x, const float* eps, const float* noise, float* x_prev,
                       float* pred_x0, int64_t nb, int64_t per, float a_t, float a_prev,
                       float sigma_t, float sqrt_one_minus_at, float cfg_scale, int cfg,
                       cs_stream_t stream);

/*
 * The same update for a CAPTURED sampling step (one HIP graph replayed for every timestep of
 * DDIMSampler.ddim_sampling, samplers/ddim.py:146-179): the step's coefficients live in device memory.
 *   cs_ddim_coefficients : HOST helper, no device work.  Fills coef5_host = {sqrt(a_t), sqrt(a_prev),
 *                          sqrt(1 - a_prev - sigma_t^2), sigma_t, sqrt_one_minus_at} with exactly the fp32
 *                          arithmetic cs_ddim_cfg_update performs, so both entry points give identical bits.
 *   cs_ddim_cfg_update_dev : coef5_dev = that block, resident on the device (the caller copies the step's
 *                          row there before each replay).  Everything else as cs_ddim_cfg_update.
 */
int cs_ddim_coefficients(float a_t, float a_prev, float sigma_t, float sqrt_one_minus_at, float* coef5_host);
int cs_ddim_cfg_update_dev(const float* x, const float* eps, const float* noise, float* x_prev,
                           float* pred_x0, int64_t nb, int64_t per, const float* coef5_dev,
                           float cfg_scale, int cfg, cs_stream_t stream);

/*
 * PLMS step (samplers/plms.py:175-236, the alternative sampler; eta must be 0 there, plms.py:29-30): fused guidance
 * combine + pseudo linear multistep combination + x0 / x_{t-1} update.
 *   e_t     = cfg ? e_uc + scale * (e_c - e_uc) : eps          -> e_out (optional: the history entry to keep)
 *   e_t'    = CS_PLMS_PLAIN      e_t
 *             CS_PLMS_AB2        (3 e_t - h1) / 2
 *             CS_PLMS_AB3        (23 e_t - 16 h1 + 5 h2) / 12
 *             CS_PLMS_AB4        (55 e_t - 59 h1 + 37 h2 - 9 h3) / 24        h1 = newest earlier prediction
 *             CS_PLMS_EULER_AVG  (h1 + e_t) / 2      second half of the start-up step: h1 = e_t of the first half,
 *                                                    eps = the model at (x_prev of the first half, t_next)
 *   pred_x0 = (x - sqrt_one_minus_at * e_t') / sqrt(a_t);   x_prev = sqrt(a_prev) * pred_x0 + sqrt(1 - a_prev) * e_t'
 * x_prev may alias x; pred_x0, e_out and unused history pointers may be NULL.
 */
#define CS_PLMS_PLAIN 0
#define CS_PLMS_AB2 1
#define CS_PLMS_AB3 2
#define CS_PLMS_AB4 3
#define CS_PLMS_EULER_AVG 4
int cs_plms_update(const float* x, const float* eps, const float* h1, const float* h2, const float* h3,
                   float* e_out, float* x_prev, float* pred_x0, int64_t nb, int64_t per, int mode, float a_t,
                   float a_prev, float sqrt_one_minus_at, float cfg_scale, int cfg, cs_stream_t stream);

/*
 * VQ nearest-code lookup (quantizer.py:76-84): z [m][ldz] (first edim entries of each row),
 * codebook [ncode][edim] -> idx[m] (int64, first minimum) and zq [m][ldq] = codebook[idx].
 * d = sum(z^2) + sum(e^2) - 2 z.e evaluated in fp32 in that order.  edim <= 4.
 */
int cs_vq_argmin_lookup(const float* z, const float* codebook, int64_t* idx, float* zq, int64_t m,
                        int ncode, int edim, int ldz, int ldq, cs_stream_t stream);

/*
 * Scene-graph convolution helpers (model/graph.py:146-151,176-199).
 *   cs_gcn_gather_cat:   out[t] = [obj[s_t] | pred[t] | obj[o_t]],  edges [t][2] int64
 *   cs_gcn_segment_mean: pooled[i] = (sum_{t: s_t = i} new_t[0:h] (edge order) then
 *                                     + sum_{t: o_t = i} new_t[off_o : off_o + h]) / max(count_i, 1)
 *     -- sequential edge order == torch CPU scatter_add order, so sums are deterministic.
 * Returns CS_EINVAL semantics cannot be reported for bad indices from the device; indices are
 * range-checked in-kernel and out-of-range edges are skipped with *err set to 1.
 */
int cs_gcn_gather_cat(const float* obj, const float* pred, const int64_t* edges, float* out,
                      int n_obj, int n_tri, int d_obj, int d_pred, int32_t* err,
                      cs_stream_t stream);
int cs_gcn_segment_mean(const float* new_t, const int64_t* edges, float* pooled, int n_obj,
                        int n_tri, int h, int off_o, int ld_t, int32_t* err, cs_stream_t stream);

/*
 * The same pooling through a CSR-by-destination index built once per graph (the lists depend on the edges only; the
 * five layers of a GraphTripleConvNet share it): O(T) per layer and channel instead of O(O T), same summation order,
 * same bits.  csr: cs_gcn_csr_ints(n_obj, n_tri) int32 of caller-owned scratch.
 */
int64_t cs_gcn_csr_ints(int n_obj, int n_tri);
int cs_gcn_csr_build(const int64_t* edges, int32_t* csr, int n_obj, int n_tri, int32_t* err, cs_stream_t stream);
int cs_gcn_segment_mean_csr(const float* new_t, const int32_t* csr, float* pooled, int n_obj, int h, int off_o,
                            int ld_t, cs_stream_t stream);

/* Embedding row gather: out[i] = table[idx[i]] (nn.Embedding, VAEGAN_V2FULL.py:225-226). */
int cs_embedding(const float* table, const int64_t* idx, float* out, int n, int dim, int n_rows,
                 int ldo, int32_t* err, cs_stream_t stream);

/* Row-wise log_softmax over [m][c] (F.log_softmax(angle_net(...), dim=1), VAEGAN_V2FULL.py:286). */
int cs_log_softmax(const float* x, float* y, int m, int c, int ldx, int ldy, cs_stream_t stream);

/*
 * Deterministic synthetic tensor fill (benchmarks / parity tests; no checkpoints are reachable offline):
 *   z = splitmix64(base + i * 0x9E3779B97F4A7C15);  k = z >> 40;
 *   out[i] = (float)( (double)(2k - 2^24) / 2^24 * scale + offset )
 * bit-identical to commonscenes_amd/synth.py::tensor (integer ops exact, two correctly-rounded fp64 ops,
 * one fp64->fp32 rounding).
 */
int cs_synth_fill(float* out, int64_t n, uint64_t base, double scale, double offset, cs_stream_t stream);

/*
 * One direction of the Chamfer distance (extension/chamfer.cu:11-75 NmDistanceKernel, bound by
 * extension/chamfer_cuda.cpp:9-26 and wrapped by extension/dist_chamfer.py:12-29; used by the diversity metric of
 * scripts/eval_3dfront.py:394-397): for each of the n points of xyz1 [b][n][3] the squared distance to, and the
 * index of, its nearest neighbour among the m points of xyz2 [b][m][3].  Ties keep the lowest index.
 * The reference's forward is this call twice (xyz1 -> xyz2, xyz2 -> xyz1).
 */
int cs_chamfer_nm_distance(const float* xyz1, const float* xyz2, float* dist, int32_t* idx, int b, int n, int m,
                           cs_stream_t stream);

/*
 * Chamfer backward (extension/chamfer.cu:155-185, bound by chamfer_cuda.cpp:28-31, dist_chamfer.py:35-47):
 *   grad_xyz1[j] = 2 g1[j] (a_j - b_idx1[j]) + sum_{k: idx2[k] == j} 2 g2[k] (a_j - b_k), and symmetrically grad_xyz2.
 * The reference scatters with atomicAdd (order-dependent rounding); this is a gather, k ascending: deterministic.
 * Outputs are written, not accumulated (the reference's wrapper passes zero-filled tensors).
 */
int cs_chamfer_backward(const float* xyz1, const float* xyz2, const float* grad_dist1, const float* grad_dist2,
                        const int32_t* idx1, const int32_t* idx2, float* grad_xyz1, float* grad_xyz2, int b, int n,
                        int m, cs_stream_t stream);

/*
 * Approximate earth mover's distance (scripts/pytorch_structural_losses/src/approxmatch.cu, bound by
 * structural_loss.cpp and wrapped by match_cost.py; used by scripts/compute_mmd_cov_1nn.py:56-62):
 *   cs_emd_approxmatch     the nine-level auction (approxmatch.cu:3-182): match [b][m][n] (query-major, as the
 *                          reference lays it out), temp [b][(n+m)*2] scratch (remainL | remainR | ratioL | ratioR)
 *   cs_emd_matchcost       out[i] = sum_k sum_j match[i][k][j] |xyz2_k - xyz1_j|          (:184-224)
 *   cs_emd_matchcost_grad  d cost / d xyz1 -> grad1 [b][n][3], d cost / d xyz2 -> grad2 [b][m][3]   (:229-320)
 * xyz1 [b][n][3], xyz2 [b][m][3] fp32.  exp is the fast hardware exponential, as in the reference (__expf).
 */
int cs_emd_approxmatch(const float* xyz1, const float* xyz2, float* match, float* temp, int b, int n, int m,
                       cs_stream_t stream);
int cs_emd_matchcost(const float* xyz1, const float* xyz2, const float* match, float* out, int b, int n, int m,
                     cs_stream_t stream);
int cs_emd_matchcost_grad(const float* xyz1, const float* xyz2, const float* match, float* grad1, float* grad2, int b,
                          int n, int m, cs_stream_t stream);

/*
 * SDF -> triangle mesh by marching cubes (model/diff_utils/util_3d.py:194-236 sdf_to_mesh, which runs
 * PyMCubes' mcubes.marching_cubes(sdf_i, level) on the CPU per object).  sdf: [nb][n][n][n] fp32 (the decoder's
 * (B,1,64,64,64) output as is), n <= 160.  A vertex per grid edge whose endpoints straddle `level` ((v < level) differs),
 * linearly interpolated in fp64 like PyMCubes; vertices unique per edge, voxel-raster order (+x, +y, +z edge of each
 * voxel); triangles from a 256-case table (commonscenes_amd/mc_tables.py), cube-raster order:
 *   table = CS_MC_TABLE_CLASSIC (0, ABI 14): the classic Lorensen-Cline table in its universally replicated 256-row form
 *     (Bourke's triTable, 820 triangles) -- the triangle SET a user of mcubes.marching_cubes gets; normals towards
 *     decreasing value, as published;
 *   table = CS_MC_TABLE_WATERTIGHT (1): derived from the cube geometry (ambiguous faces cut off the inside corners);
 *     normals towards increasing value.
 * Both mesh the same watertight surface -- same patch boundaries on all 256 cases (tests/test_mesh.py) -- and differ in
 * how polygons are fanned and in winding.  Deterministic, no atomics.
 *   cs_mc_blocks_per_object(n)  B = ceil(n^3 / 4096)
 *   cs_mc_count   -> block_sums [nb][B][2] int32: vertices and triangles per 4096-voxel block.  The caller reads them
 *                  back (the library never synchronises), sizes the outputs and passes per-object bases:
 *   cs_mc_emit    vert_base / face_base [nb] int64 (device): first vertex / triangle of each object in `verts`
 *                  [total][3] fp32 and `faces` [total][3] int64 (vertex ids local to the object);
 *                  voxel_ws: nb * n^3 int32 scratch;  vertex = index_coordinate / vert_div + vert_shift
 *                  (n, -0.5 reproduce util_3d.py:218; 1, 0 give PyMCubes' raw index coordinates).
 */
int cs_mc_blocks_per_object(int n);
#define CS_MC_TABLE_CLASSIC 0
#define CS_MC_TABLE_WATERTIGHT 1
int cs_mc_count(const float* sdf, int nb, int n, float level, int table, int32_t* block_sums, cs_stream_t stream);
int cs_mc_emit(const float* sdf, int nb, int n, float level, int table, const int32_t* block_sums, const int64_t* vert_base,
               const int64_t* face_base, float* verts, int64_t* faces, int32_t* voxel_ws, float vert_div,
               float vert_shift, cs_stream_t stream);

/*
 * Whole-forward driver (SURVEY 8b "cs_unet_step"): UNet3DModel.forward (openai_model_3d.py:752-789) with the
 * crossattn conditioning of DiffusionUNet.forward (network.py:28-30) as ONE call over a packed weight arena.
 * The plan object is host memory only (architecture walk, arena layout, packing recipe); every device buffer --
 * raw parameters, arena, context vectors, workspace -- is caller-owned, and nothing synchronises except
 * cs_unet_pack (one stream sync at load time to read the per-tensor |w| maxima in CS_MATH_F16X3).
 *
 *   cs_unet_create        config -> plan.  Scope: the two shipped families -- dims=3 with one transformer block per
 *                         SpatialTransformer3D and ONE context token (attention.py:170-199 over a single key is
 *                         the row vector to_out(to_v(ctx))), and dims=4 with AttentionBlock blocks.
 *   cs_unet_param_*       the reference state_dict entries (names as in SURVEY App. C, reference shapes) and
 *                         where each one goes in the caller's raw fp32 parameter buffer (cs_unet_raw_bytes).
 *   cs_unet_pack          raw parameters -> arena (cs_unet_arena_bytes): GEMM weights in the layout of the
 *                         plan's math mode (q|k|v fused, all ResBlock emb_layers fused, GEGLU columns
 *                         interleaved per 224-column tile), biases, norm affine parameters.
 *   cs_unet_context       once per sampling run: ctx [nb_ctx][context_dim] -> ctxvec [nb_ctx][cs_unet_context_floats]
 *   status                (cs_unet_context / cs_unet_step / cs_vqvae_decode) sticky CS_STATUS_* word on the device, or NULL.
 *   cs_unet_step          eps = UNet(x, t, ctx).  x: NCDHW [nb_x][in_channels][d][h][w]; t: int64 [nb_x];
 *                         cfg_pairs = 0: nb_ctx == nb_x.  cfg_pairs = 1 (classifier-free guidance, ddim.py:206-209):
 *                         the SAME (x, t) under two contexts, ctxvec holds 2*nb_x rows [uc; c], the context-free
 *                         prefix runs once, eps is [2*nb_x] = [eps_uc; eps_c].  workspace >= cs_unet_workspace_bytes.
 * All return CS_OK / CS_EINVAL / CS_ENOMEM (workspace too small) / a hipError_t.
 */
typedef struct CsUnetConfig {
  int32_t in_channels, out_channels, model_channels, num_res_blocks;
  int32_t n_mult;
  int32_t channel_mult[8];
  int32_t n_attn_res;
  int32_t attention_resolutions[8];
  int32_t num_heads, context_dim;
  int32_t d, h, w;   /* latent grid, 16 x 16 x 16 */
  int32_t math;      /* CS_MATH_FP32 / CS_MATH_F16X3 */
  /* 1: SpatialTransformer3D blocks + one-token context (config/sdfusion-txt2shape.yaml, crossattn conditioning);
   * 0: AttentionBlock / QKVAttentionLegacy blocks, no context (config/sdfusion-txt2shape_concat.yaml: the condition
   *    volume is x's last input channel, cs_unet_step then takes ctxvec = NULL and cfg_pairs = 0). */
  int32_t use_spatial_transformer;
  int32_t dims;      /* 3: Down/Upsample act on H, W only; 4: on D, H, W (openai_model_3d.py:150-155,188) */
} CsUnetConfig;
typedef struct cs_unet cs_unet;

int cs_unet_create(const CsUnetConfig* cfg, cs_unet** out);
void cs_unet_destroy(cs_unet* u);
int cs_unet_param_count(const cs_unet* u);
int cs_unet_param_info(const cs_unet* u, int i, const char** name, int64_t shape5[5], int* ndim,
                       int64_t* raw_offset_bytes);
int64_t cs_unet_raw_bytes(const cs_unet* u);
int64_t cs_unet_arena_bytes(const cs_unet* u);
int64_t cs_unet_context_floats(const cs_unet* u);
/* (ABI 15, r5) the largest |entry| of every transformer block's cross-attention row vector of the CURRENT run -- block order:
 * input blocks, middle block, output blocks; block b's columns of cs_unet_context's output are the next c_b floats -- read
 * back by the host once per sampling run and handed over here: it enters the static bound of the block's t1 / t2 operands
 * (cs_transformer_static_scales).  Never called: 0 is assumed (a run whose context vectors are large may then raise
 * CS_STATUS_F16X3_OVERFLOW on proj_out's operand, as before r5).  Host-only. */
int cs_unet_set_context_bounds(cs_unet* u, const float* ctx_max, int n_blocks);
int cs_unet_pack(cs_unet* u, const void* raw_dev, void* arena_dev, cs_stream_t stream);
int64_t cs_unet_workspace_bytes(const cs_unet* u, int nb_x, int cfg_pairs);
int cs_unet_context(const cs_unet* u, const void* arena, const float* ctx, int nb_ctx, float* ctxvec,
                    int32_t* status, void* workspace, int64_t workspace_bytes, cs_stream_t stream);
int cs_unet_step(const cs_unet* u, const void* arena, const float* x_ncdhw, const int64_t* t,
                 const float* ctxvec, float* eps_ncdhw, int nb_x, int cfg_pairs, int32_t* status, void* workspace,
                 int64_t workspace_bytes, cs_stream_t stream);

/*
 * Whole-decode driver: VQVAE.decode_no_quant / decode (vqvae_networks/network.py:90-103) -- nearest-code
 * quantisation (quantizer.py:76-84), post_quant_conv, Decoder3D (vqvae_modules.py:292-409) -- as ONE call over a
 * packed weight arena; same ownership rules as cs_unet_* (host-only plan, caller-owned raw / arena / workspace).
 *   latent_ncdhw : [nb][embed_dim][g][g][g], g = resolution >> (n_mult - 1)   (3 x 16^3 for config/vqvae_snet.yaml)
 *   sdf_ncdhw    : [nb][out_ch][resolution]^3
 *   quantize     : 1 = decode_no_quant(h) (despite its name the reference quantises first, network.py:97-101);
 *                  0 = decode(quant) / force_not_quantize.  code_indices: int64 [nb * g^3] or NULL.
 * The parameter table lists the decode-side state_dict entries of the reference VQVAE (decoder.*,
 * quantize.embedding.weight, post_quant_conv.*).
 */
typedef struct CsVqvaeConfig {
  int32_t ch, out_ch;
  int32_t n_mult;
  int32_t ch_mult[8];
  int32_t num_res_blocks, z_channels, resolution;
  int32_t n_embed, embed_dim;
  int32_t math;
} CsVqvaeConfig;
typedef struct cs_vqvae cs_vqvae;

int cs_vqvae_create(const CsVqvaeConfig* cfg, cs_vqvae** out);
void cs_vqvae_destroy(cs_vqvae* u);
int cs_vqvae_param_count(const cs_vqvae* u);
int cs_vqvae_param_info(const cs_vqvae* u, int i, const char** name, int64_t shape5[5], int* ndim,
                        int64_t* raw_offset_bytes);
int64_t cs_vqvae_raw_bytes(const cs_vqvae* u);
int64_t cs_vqvae_arena_bytes(const cs_vqvae* u);
int cs_vqvae_pack(cs_vqvae* u, const void* raw_dev, void* arena_dev, cs_stream_t stream);
int64_t cs_vqvae_workspace_bytes(const cs_vqvae* u, int nb);
int cs_vqvae_decode(const cs_vqvae* u, const void* arena, const float* latent_ncdhw, float* sdf_ncdhw,
                    int64_t* code_indices, int nb, int quantize, int32_t* status, void* workspace,
                    int64_t workspace_bytes, cs_stream_t stream);

/* Library / device self-description. */
int cs_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* COMMONSCENES_HIP_H */
