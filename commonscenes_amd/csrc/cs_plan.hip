// Host-side decisions that every host of the library shares (r4, VERDICT r3 next #5: "one plan, one place").
//
// Until r3 the Python host (ops.py) and the native drivers (cs_driver.h) each carried their own copy of the rules that
// decide HOW an operator is run -- the F16X3 operand scale of a normalisation-fed GEMM, whether a GroupNorm emits the
// pre-split operand pair for the conv that follows, the tile of a taps-as-columns GEMM, the row threshold of the
// channel-split ResBlocks -- and ~40 CS_* environment switches were parsed wherever they were used, in both languages.
// The rules live here now, once, behind the C ABI: both hosts call them (and cs_conv_gemm_plan / cs_conv_gemm_epilogue_caps /
// cs_conv_gemm_launch_info in cs_gemm.hip), and the switches are ONE struct, CsDebug, parsed ONCE from the environment.
// Nothing in this file touches the device.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "cs_common.h"

namespace {

CsDebug g_dbg;
std::once_flag g_once;

bool flag(const char* name) {       // set, non-empty and not "0"
  const char* e = getenv(name);
  return e && *e && !(e[0] == '0' && e[1] == 0);
}
long long num(const char* name, long long dflt) {
  const char* e = getenv(name);
  return (e && *e) ? atoll(e) : dflt;
}

void parse_env(CsDebug& d) {
  memset(&d, 0, sizeof(d));
  d.no_split16 = flag("CS_NO_SPLIT16");
  d.split16_min_rows = num("CS_SPLIT16_MIN_ROWS", 8192);
  d.no_pair16 = flag("CS_NO_PAIR16");
  d.no_upfold = flag("CS_NO_UPFOLD");
  d.no_splitk = flag("CS_NO_SPLITK");
  d.no_fused_geglu = flag("CS_NO_FUSED_GEGLU");
  d.no_tapcol = flag("CS_NO_TAPCOL");
  d.tapcol_tile = (int32_t)num("CS_TAPCOL_TILE", 0);
  d.no_cfg_split = flag("CS_NO_CFG_SPLIT");
  d.cfg_split_min_rows = num("CS_CFG_SPLIT_MIN_ROWS", 65536);
  d.concat_copy = flag("CS_CONCAT_COPY");
  d.gn_small_group = num("CS_GN_SMALL_GROUP", 11264);
  d.tile512 = flag("CS_TILE512");
  d.no_pw = flag("CS_NO_PW");
  d.no_slab4 = flag("CS_NO_SLAB4");
  d.no_attn_img = flag("CS_NO_ATTN_IMG");
  d.attn_nw8 = flag("CS_ATTN_NW8");
  d.no_up2_direct = flag("CS_NO_UP2_DIRECT");
  d.no_up2_batch = flag("CS_NO_UP2_BATCH");
  d.plan_pow2 = flag("CS_PLAN_POW2");
  d.slice_tile2 = flag("CS_SLICE_TILE2");
  d.no_gn_parts = flag("CS_NO_GN_PARTS");
  d.no_pair_epilogue = flag("CS_NO_PAIR_EPILOGUE");
  d.no_dyn_scale = flag("CS_NO_DYN_SCALE");
  d.no_tok_rules = flag("CS_NO_TOK_RULES");
  d.no_fused_reduce = flag("CS_NO_FUSED_REDUCE");
  d.no_gn_fold = flag("CS_NO_GN_FOLD");
  d.no_kwave = flag("CS_NO_KWAVE");
  d.no_static_scales = flag("CS_NO_STATIC_SCALES");
  d.no_wino = flag("CS_NO_WINO");
  d.wino_min_rows = (int32_t)num("CS_WINO_MIN_ROWS", 1024);
  d.no_wino43 = flag("CS_NO_WINO43");
  d.wino43_min_rows = (int32_t)num("CS_WINO43_MIN_ROWS", 2048);
  d.no_wino_tail = flag("CS_NO_WINO_TAIL");
}

}  // namespace

extern "C" const CsDebug* cs_debug(void) {
  std::call_once(g_once, [] { parse_env(g_dbg); });
  return &g_dbg;
}

extern "C" void cs_debug_set(const CsDebug* d) {
  (void)cs_debug();
  if (d)
    g_dbg = *d;
  else
    parse_env(g_dbg);
}

// F16X3 operand scale of a GEMM fed by a GroupNorm / LayerNorm (+ SiLU / GELU / identity) taking its statistics over n
// elements.  A normalised value obeys |x^| <= sqrt(n - 1), so |y| <= gmax * sqrt(n - 1) + bmax =: bound (|silu(y)|,
// |gelu(y)| <= |y|): the largest power of two 2^k with bound * 2^k <= 65000 < 65504 cannot leave the fp16 range WHATEVER the
// input -- the producer's bound replaces a fixed guess (r3) -- and is 16-128x larger than the raw-activation default 16 for
// the shipped layers, so the absolute floor 2^-25 / a_scale of tiny operands drops accordingly.  k is clamped to [-8, 40].
extern "C" float cs_norm_a_scale(float gmax, float bmax, int64_t n) {
  const double bound = (double)gmax * std::sqrt((double)(n > 1 ? n - 1 : 1)) + (double)bmax;
  if (!(bound > 0.0) || !std::isfinite(bound)) return (float)std::ldexp(1.0, 40);
  int ex = 0;
  (void)std::frexp(65000.0 / bound, &ex);
  int k = ex - 1;
  if (k < -8) k = -8;
  if (k > 40) k = 40;
  return (float)std::ldexp(1.0, k);
}

extern "C" float cs_bound_a_scale(float bound) {
  if (!(bound > 0.f) || !std::isfinite(bound)) return (float)std::ldexp(1.0, 40);
  int ex = 0;
  (void)std::frexp(65000.0 / (double)bound, &ex);
  int k = ex - 1;
  if (k < -24) k = -24;
  if (k > 40) k = 40;
  return (float)std::ldexp(1.0, k);
}

// r6: the same for an attention block fed by a GroupNorm -- see the header.  ONE rule for both hosts
// (vqvae.py::_attn / cs_vqvae.hip, unet.py::_attnblock / cs_unet.hip::attnblock).
extern "C" int cs_attnblock_static_scales(float gmax, float bmax, int64_t n, int c, float w_l2max, float b_absmax, float qk_scale,
                                          float* out4) {
  if (!out4 || n < 1 || c < 1 || !(qk_scale > 0.f) || !(gmax >= 0.f) || !(bmax >= 0.f) || !(w_l2max >= 0.f) || !(b_absmax >= 0.f))
    return CS_EINVAL;
  const double e = (double)gmax * std::sqrt((double)(n > 1 ? n - 1 : 1)) + (double)bmax;
  const double b = (double)w_l2max * std::sqrt((double)c) * e + (double)b_absmax;
  auto S = [](double v) { return cs_bound_a_scale(v < 3.0e38 ? (float)(v * (1.0 + 1e-6)) : 3.0e38f); };
  out4[0] = S(b * (double)qk_scale);
  out4[1] = S(b);
  out4[2] = S(b);
  out4[3] = S(b);
  return CS_OK;
}

// r5: F16X3 operand scales of the operands BORN INSIDE a transformer block (attention.py:237-245,
// 335-351) from bounds that hold for every input -- so they can never leave the fp16 range and CS_STATUS_F16X3_OVERFLOW is a
// pure assertion there.  ONE rule for both hosts (unet.py::_static_scales, cs_unet.hip::attn_block).
//   st (CsTransformerStats): max row 2-norms r* and max |bias| b* of the block's Linears (cs_weight_rowstats), max |gamma|
//   and ||beta||_2 of LayerNorm 1 / 3;  gn_gmax / gn_bmax: max |gamma| / |beta| of the SpatialTransformer's GroupNorm;
//   ctx_max: the largest |entry| of the block's one-token cross-attention row vector (data of the RUN, read once per run).
//     Y1 = g1 sqrt(c) + ||b1||            >= ||LayerNorm1(x) token||_2   (a normalised token has 2-norm sqrt(c))
//     |q| <= rq Y1, |k| <= rk Y1, |v| <= rv Y1 =: Bv  (no bias);   |attention output| <= Bv  (softmax rows: convex weights)
//     |t0| <= rpi sqrt(c) (gn_gmax sqrt(n - 1) + gn_bmax) + bpi,   n = tokens * c / 32 elements per GroupNorm group
//     |t1| <= ro sqrt(c) Bv + bo + |t0| + ctx_max
//     |x|, |gate| of the GEGLU <= rx Y3 + bx, rg Y3 + bg;   |gg| <= their product  (|gelu(g)| <= |g|)
//     |t2| <= r2 sqrt(4c) |gg| + b2 + |t1|
//   out[0..2] = scales of q * dh^-1/2, k, v (cs_attn_selfattn_f16x3_scaled), out[3] = attention output -> to_out,
//   out[4] = gg -> ff.net.2, out[5] = t2 -> proj_out;  out[6..11] = the bounds (q, k, v, t1, gg, t2), for reports.
// The 2-norm chains overshoot by one to three orders of magnitude: with scale = 65000 / bound an operand's absolute floor
// is 2^-25 bound / 65000 ~ 5e-13 bound, still fp32 grade for values five orders of magnitude below the bound.
extern "C" int cs_transformer_static_scales(const CsTransformerStats* st, int c, int64_t n_tokens, int heads, float gn_gmax,
                                            float gn_bmax, float ctx_max, float* out12) {
  if (!st || !out12 || c <= 0 || heads <= 0 || c % heads || n_tokens <= 0) return CS_EINVAL;
  const double rc = std::sqrt((double)c);
  const double dh = (double)(c / heads);
  const double y1 = (double)st->g1 * rc + (double)st->be1;
  const double bq = (double)st->rq * y1, bk = (double)st->rk * y1, bv = (double)st->rv * y1;
  const double ngrp = (double)n_tokens * (double)(c / 32 > 0 ? c / 32 : 1);
  const double egn = (double)gn_gmax * std::sqrt(ngrp > 1.0 ? ngrp - 1.0 : 1.0) + (double)gn_bmax;
  const double bt0 = (double)st->rpi * rc * egn + (double)st->bpi;
  const double bt1 = (double)st->ro * rc * bv + (double)st->bo + bt0 + (double)(ctx_max > 0.f ? ctx_max : 0.f);
  const double y3 = (double)st->g3 * rc + (double)st->be3;
  const double bgg = ((double)st->rx * y3 + (double)st->bx) * ((double)st->rg * y3 + (double)st->bg);
  const double bt2 = (double)st->r2 * std::sqrt(4.0 * c) * bgg + (double)st->b2 + bt1;
  auto S = [](double b) { return cs_bound_a_scale(b < 3.0e38 ? (float)(b * (1.0 + 1e-6)) : 3.0e38f); };
  out12[0] = S(bq / std::sqrt(dh));
  out12[1] = S(bk);
  out12[2] = S(bv);
  out12[3] = S(bv);
  out12[4] = S(bgg);
  out12[5] = S(bt2);
  out12[6] = (float)bq; out12[7] = (float)bk; out12[8] = (float)bv; out12[9] = (float)bt1; out12[10] = (float)bgg;
  out12[11] = (float)bt2;
  return CS_OK;
}

// Should the GroupNorm feeding a conv (cout x cin x k^3, `plain`: neither a folded Upsample conv nor taps-as-columns) over m
// output rows emit the fp16 hi / lo operand pair (CsConvGemm.a_format = 1)?  Yes where that conv runs the slab kernel on a
// 256-row tile -- there the in-loop conversion is what is left to remove -- and, for the 224-column convs of
// medium batches, on the 128-row slab tile from split16_min_rows rows (r3: 7 objects 27.34 -> 26.94 ms/step; below it
// slower).  Bit-identical to the fp32 route either way.
extern "C" int cs_conv_wants_split16(int64_t m, int cin, int cout, int k, int plain, int math) {
  const CsDebug* d = cs_debug();
  if (d->no_split16 || math != CS_MATH_F16X3 || !plain || k != 3 || (cin & 7)) return 0;
  const int64_t t256 = (m + 255) / 256;
  if (cout % 224 == 0) return t256 * (cout / 224) >= 192 || (d->split16_min_rows > 0 && m >= d->split16_min_rows);
  if (cout % 128 == 0) return t256 * (cout / 128) >= 192;
  return (cout == 64 || cout <= 4) && t256 >= 192;
}

// Tile of the pointwise GEMM of a taps-as-columns conv (cs_pack_weight_f16x3_tapcol: ncolp = 27 * cout + pad columns):
// 256-row tiles once they fill the chip, the library's automatic choice (0) below.
extern "C" int cs_tapcol_tile(int64_t m, int ncolp) {
  const CsDebug* d = cs_debug();
  if (d->tapcol_tile) return d->tapcol_tile;
  if ((m + 255) / 256 < 192) return 0;
  return ncolp <= 64 ? 7 : 6;
}

// Is a 3x3x3 conv with this few output channels run as "taps as columns" ?  The callers name the layers
// (UNet `out.2`, VQ decoder `conv_out`); this is the rule they are then held to.
extern "C" int cs_tapcol_ok(int cout, int cin, int k, int math) {
  return !cs_debug()->no_tapcol && math == CS_MATH_F16X3 && k == 3 && cout <= 4 && (cin & 3) == 0;
}
