// cs_unet_*: the whole UNet forward of the shape-branch denoiser as ONE native call over a packed weight arena
// (SURVEY 8b "cs_unet_step").  Host-side C++ only sequences the library's own kernels (cs_conv_gemm,
// cs_groupnorm_*, cs_layernorm, cs_attn_selfattn*, cs_copy_rows, ...) -- the same launches, in the same order,
// with the same arguments as commonscenes_amd/unet.py::DiffusionUNet.forward_ndhwc, so both drivers produce
// identical bits -- and owns nothing on the device: the caller passes the raw parameters, the arena, the
// workspace and the stream.
//
// Reference being replaced:
//   model/networks/diffusion_networks/openai_model_3d.py:452-789   UNet3DModel.__init__ / forward
//   model/networks/diffusion_networks/attention.py:154-351         CrossAttention / BasicTransformerBlock /
//                                                                  SpatialTransformer3D
//   model/networks/diffusion_networks/network.py:20-42             DiffusionUNet.forward (crossattn branch)
// Scope: the shipped config family (dims=3 i.e. H,W-only resampling, use_spatial_transformer, one transformer
// block per SpatialTransformer3D, ONE context token -- SURVEY F4: cross-attention over a single key is the
// per-sample row vector to_out(to_v(ctx)), computed once per sampling run by cs_unet_context).
#include "cs_driver.h"

namespace {

enum Kind { CONV_IN, RES, ATTN, ATTNBLOCK, DOWN, UP };

struct Layer {
  Kind kind;
  int cin, cout;
  int g[8];
  int n[4];
  int emb_lo = 0;      // RES: column offset into the batched emb_layers projection
  // RES of an output block whose skip half is shared by the guidance halves (unet.py::_pack "channel-split ResBlocks"):
  // split channel ks (> 0), channels of h, and the GEMMs {in_layers.2 h, in_layers.2 s, skip_connection h, skip_connection s}
  int ks = 0, ch_h = 0;
  int gsp[4] = {-1, -1, -1, -1};
  int ctx_off = 0;     // ATTN: column offset into the context-vector block
  bool fused_geglu = false;
  // ATTN, F16X3 (r5): what bounds the operands born inside the transformer block (cs_transformer_static_scales): filled by
  // cs_unet_pack from the raw weights; ctx_max = the largest |entry| of the block's cross-attention row vector of the
  // current run (cs_unet_set_context_bounds), 0 until the host sets it
  CsTransformerStats ts;
  bool has_ts = false;
  float ctx_max = 0.f;
  int ts_slot = -1;    // first of this block's 34 float slots behind the plan's |w| maxima
  // ATTNBLOCK, F16X3 (r6): max row 2-norm of the fused qkv weight and max |bias| (cs_attnblock_static_scales), filled by
  // cs_unet_pack; ab_slot = the block's 4 float slots, ab_w / ab_b = the raw parameters
  int ab_slot = -1, ab_w = -1, ab_b = -1;
  float ab_l2max = 0.f, ab_bmax = 0.f;
};

}  // namespace

struct cs_unet : Plan {
  CsUnetConfig cfg;
  std::vector<std::vector<Layer>> inp, out;
  std::vector<Layer> mid;
  int g_te0 = -1, g_te2 = -1, g_emb_all = -1, g_out = -1, n_out = -1;
  int emb_total = 0, ctx_total = 0, cpad_in = 4, ch_final = 0;
  int64_t split_min_rows = 65536;      // channel-split ResBlocks only from this many rows (unet.py: split_min_rows)
};

namespace {

// ---------------------------------------------------------------------------------------------------------
// plan construction (mirrors unet.py::unet_blocks / unet_param_shapes / DiffusionUNet._pack)
// ---------------------------------------------------------------------------------------------------------
struct EmbAcc {
  std::vector<Piece> w, b;
  int total = 0;
};

Layer make_res(cs_unet& u, const std::string& p, int cin, int cout, int ted, EmbAcc& emb) {
  Layer l{};
  l.kind = RES;
  l.cin = cin;
  l.cout = cout;
  l.n[0] = add_norm(u, p + ".in_layers.0", cin);
  l.g[0] = add_layer_gemm(u, p + ".in_layers.2", cout, cin, 3);
  int wp, bp;
  add_wb(u, p + ".emb_layers.1", cout, ted, 0, true, wp, bp);
  emb.w.push_back({wp, 0, cout});
  emb.b.push_back({bp, 0, cout});
  l.emb_lo = emb.total;
  emb.total += cout;
  l.n[1] = add_norm(u, p + ".out_layers.0", cout);
  l.g[1] = add_layer_gemm(u, p + ".out_layers.3", cout, cout, 3);
  l.g[2] = (cin != cout) ? add_layer_gemm(u, p + ".skip_connection", cout, cin, 1) : -1;
  return l;
}

Layer make_attn(cs_unet& u, const std::string& p, int c, int ctx_dim, bool f16x3) {
  Layer l{};
  l.kind = ATTN;
  l.cin = l.cout = c;
  l.n[0] = add_norm(u, p + ".norm", c);
  l.g[0] = add_layer_gemm(u, p + ".proj_in", c, c, 1);
  const std::string t = p + ".transformer_blocks.0";
  int q, k, v, dummy;
  add_wb(u, t + ".attn1.to_q", c, c, 0, false, q, dummy);
  add_wb(u, t + ".attn1.to_k", c, c, 0, false, k, dummy);
  add_wb(u, t + ".attn1.to_v", c, c, 0, false, v, dummy);
  l.g[1] = add_gemm(u, {{q, 0, c}, {k, 0, c}, {v, 0, c}}, {}, 3 * c, c, 0);      // fused q|k|v projection
  l.g[2] = add_layer_gemm(u, t + ".attn1.to_out.0", c, c, 0);
  // attn2.to_q / to_k are parameters of the reference module but unused with one context token (softmax == 1)
  add_wb(u, t + ".attn2.to_q", c, c, 0, false, q, dummy);
  add_wb(u, t + ".attn2.to_k", c, ctx_dim, 0, false, k, dummy);
  l.g[3] = add_layer_gemm(u, t + ".attn2.to_v", c, ctx_dim, 0, false);
  l.g[4] = add_layer_gemm(u, t + ".attn2.to_out.0", c, c, 0);
  int wp, bp;
  add_wb(u, t + ".ff.net.0.proj", 8 * c, c, 0, true, wp, bp);
  const int hdim = 4 * c;
  l.fused_geglu = f16x3 && (hdim % 112 == 0);
  if (l.fused_geglu) {
    // output columns interleaved per 224-column tile as [x (112) | gate (112)]  (ops.pack_geglu_weight)
    std::vector<Piece> w, b;
    for (int blk = 0; blk < hdim / 112; ++blk) {
      w.push_back({wp, blk * 112, 112});
      w.push_back({wp, hdim + blk * 112, 112});
      b.push_back({bp, blk * 112, 112});
      b.push_back({bp, hdim + blk * 112, 112});
    }
    l.g[5] = add_gemm(u, w, b, 8 * c, c, 0);
  } else {
    l.g[5] = add_gemm(u, {{wp, 0, 8 * c}}, {{bp, 0, 8 * c}}, 8 * c, c, 0);
  }
  l.g[6] = add_layer_gemm(u, t + ".ff.net.2", c, 4 * c, 0);
  l.n[1] = add_norm(u, t + ".norm1", c);
  l.n[2] = add_norm(u, t + ".norm2", c);
  l.n[3] = add_norm(u, t + ".norm3", c);
  l.g[7] = add_layer_gemm(u, p + ".proj_out", c, c, 1);
  l.ctx_off = u.ctx_total;
  u.ctx_total += c;
  return l;
}

// AttentionBlock (openai_model_3d.py:316-366; the concat family, use_spatial_transformer=False): GroupNorm ->
// Conv1d(k=1) qkv -> QKVAttentionLegacy -> Conv1d(k=1) proj_out + x.  The legacy module reads the 3C qkv channels
// as [head][q|k|v][ch]; the packed weight's rows are gathered as [q|k|v][head][ch] instead, so q, k, v are plain
// column slices for the flash kernel (same permutation as unet.py::_pack).
Layer make_attnblock(cs_unet& u, const std::string& p, int c, int heads) {
  Layer l{};
  l.kind = ATTNBLOCK;
  l.cin = l.cout = c;
  l.n[0] = add_norm(u, p + ".norm", c);
  const int wq = add_param(u, p + ".qkv.weight", {3 * c, c, 1});
  const int bq = add_param(u, p + ".qkv.bias", {3 * c});
  const int ch = c / heads;
  std::vector<Piece> w, b;
  for (int j = 0; j < 3; ++j)
    for (int h = 0; h < heads; ++h) {
      w.push_back({wq, h * 3 * ch + j * ch, ch});
      b.push_back({bq, h * 3 * ch + j * ch, ch});
    }
  l.g[0] = add_gemm(u, w, b, 3 * c, c, 0);
  l.ab_w = wq;
  l.ab_b = bq;
  const int wo = add_param(u, p + ".proj_out.weight", {c, c, 1});
  const int bo = add_param(u, p + ".proj_out.bias", {c});
  l.g[1] = add_gemm(u, {{wo, 0, c}}, {{bo, 0, c}}, c, c, 0);
  return l;
}

bool in_list(const int32_t* v, int n, int x) {
  for (int i = 0; i < n; ++i)
    if (v[i] == x) return true;
  return false;
}

int build(cs_unet& u) {
  const CsUnetConfig& c = u.cfg;
  if (c.model_channels <= 0 || c.model_channels % 32 || c.num_res_blocks <= 0 || c.n_mult <= 0 || c.n_mult > 8 ||
      c.n_attn_res < 0 || c.n_attn_res > 8 || c.num_heads <= 0 || c.in_channels <= 0 ||
      (c.use_spatial_transformer && c.context_dim <= 0) || (c.dims != 3 && c.dims != 4) ||
      c.out_channels <= 0 || c.d <= 0 || c.h <= 0 || c.w <= 0)
    return CS_EINVAL;
  if (c.math != CS_MATH_FP32 && c.math != CS_MATH_F16X3) return CS_EINVAL;
  if ((c.h >> (c.n_mult - 1)) << (c.n_mult - 1) != c.h || (c.w >> (c.n_mult - 1)) << (c.n_mult - 1) != c.w)
    return CS_EINVAL;
  if (c.dims == 4 && (c.d >> (c.n_mult - 1)) << (c.n_mult - 1) != c.d) return CS_EINVAL;
  auto attn_layer = [&](const std::string& q, int chn) {
    return c.use_spatial_transformer ? make_attn(u, q, chn, c.context_dim, c.math == CS_MATH_F16X3)
                                     : make_attnblock(u, q, chn, c.num_heads);
  };
  const int mc = c.model_channels, ted = 4 * mc, nres = c.num_res_blocks;
  const std::string P = "diffusion_net.";
  u.cpad_in = (c.in_channels + 3) / 4 * 4;
  u.g_te0 = add_layer_gemm(u, P + "time_embed.0", ted, mc, 0);
  u.g_te2 = add_layer_gemm(u, P + "time_embed.2", ted, ted, 0);
  EmbAcc emb;
  std::vector<int> chans;
  int ch = mc, ds = 1, bi = 0;
  {
    Layer l{};
    l.kind = CONV_IN;
    l.cin = c.in_channels;
    l.cout = mc;
    l.g[0] = add_layer_gemm(u, P + "input_blocks.0.0", mc, c.in_channels, 3, true, u.cpad_in);
    u.inp.push_back({l});
    chans.push_back(mc);
    bi = 1;
  }
  for (int level = 0; level < c.n_mult; ++level) {
    const int m = c.channel_mult[level];
    if (m <= 0) return CS_EINVAL;
    for (int r = 0; r < nres; ++r) {
      const std::string bp = P + "input_blocks." + std::to_string(bi++);
      std::vector<Layer> layers;
      layers.push_back(make_res(u, bp + ".0", ch, m * mc, ted, emb));
      ch = m * mc;
      if (ch % c.num_heads) return CS_EINVAL;
      if (in_list(c.attention_resolutions, c.n_attn_res, ds))
        layers.push_back(attn_layer(bp + ".1", ch));
      u.inp.push_back(layers);
      chans.push_back(ch);
    }
    if (level != c.n_mult - 1) {
      const std::string bp = P + "input_blocks." + std::to_string(bi++);
      Layer l{};
      l.kind = DOWN;
      l.cin = l.cout = ch;
      l.g[0] = add_layer_gemm(u, bp + ".0.op", ch, ch, 3);
      u.inp.push_back({l});
      chans.push_back(ch);
      ds *= 2;
    }
  }
  u.mid.push_back(make_res(u, P + "middle_block.0", ch, ch, ted, emb));
  u.mid.push_back(attn_layer(P + "middle_block.1", ch));
  u.mid.push_back(make_res(u, P + "middle_block.2", ch, ch, ted, emb));
  int oi = 0;
  for (int level = c.n_mult - 1; level >= 0; --level) {
    const int m = c.channel_mult[level];
    for (int i = 0; i <= nres; ++i) {
      const int ich = chans.back();
      chans.pop_back();
      const std::string bp = P + "output_blocks." + std::to_string(oi++);
      std::vector<Layer> layers;
      layers.push_back(make_res(u, bp + ".0", ch + ich, mc * m, ted, emb));
      ch = mc * m;
      if (in_list(c.attention_resolutions, c.n_attn_res, ds))
        layers.push_back(attn_layer(bp + ".1", ch));
      if (level && i == nres) {
        Layer l{};
        l.kind = UP;
        l.cin = l.cout = ch;
        l.g[0] = add_layer_gemm(u, bp + "." + std::to_string(layers.size()) + ".conv", ch, ch, 3, true, 0,
                                c.dims == 3 ? 3 : 7);      // Upsample: H, W doubled (dims = 3) or D, H, W
        layers.push_back(l);
        ds /= 2;
      }
      u.out.push_back(layers);
    }
  }
  // channel-split ResBlocks: output block j reads the skip of input block (n_in - 1 - j); the ones from the context-free
  // prefix of the input path (before the first attention block) are shared by the guidance halves.  Same rule and same
  // split point as unet.py::_pack.
  if (c.use_spatial_transformer && !cs_debug()->no_cfg_split) {
    size_t n_prefix = 0;
    while (n_prefix < u.inp.size()) {
      bool attn = false;
      for (const Layer& l : u.inp[n_prefix]) attn |= (l.kind == ATTN || l.kind == ATTNBLOCK);
      if (attn) break;
      ++n_prefix;
    }
    for (size_t j = 0; j < u.out.size(); ++j) {
      const size_t src = u.inp.size() - 1 - j;
      Layer& l = u.out[j][0];
      if (src >= n_prefix || l.kind != RES || l.g[2] < 0) continue;
      const int ch_s = u.inp[src].back().cout, C = l.cin, ch_h = C - ch_s, cpg = C / 32;
      if (C % 32) continue;
      int ks = 0;
      for (int k = (ch_h + 15) / 16 * 16; k < C; k += 16)
        if ((k / cpg) * cpg >= ch_h) {
          ks = k;
          break;
        }
      if (!ks || (C - ks) % 16) continue;
      for (int which = 0; which < 1; ++which) {          // in_layers.2 (3x3x3); the 1x1x1 skip_connection stays whole
        const int gi = l.g[which == 0 ? 0 : 2];
        const int wp = u.gemms[gi].w[0].param, bp = u.gemms[gi].b.empty() ? -1 : u.gemms[gi].b[0].param;
        const int k = u.gemms[gi].k;     // (the unsplit GEMM stays packed: small batches take it, see split_min_rows)
        l.gsp[2 * which] = add_gemm_cin_range(u, wp, bp, l.cout, C, k, 0, ks);
        l.gsp[2 * which + 1] = add_gemm_cin_range(u, wp, -1, l.cout, C, k, ks, C);
      }
      l.ks = ks;
      l.ch_h = ch_h;
    }
  }
  u.split_min_rows = cs_debug()->cfg_split_min_rows;
  u.ch_final = ch;
  u.n_out = add_norm(u, P + "out.0", ch);
  u.g_out = add_layer_gemm(u, P + "out.2", c.out_channels, mc, 3, true, 0, 0, /*tapcol=*/true);
  // all ResBlock emb_layers Linears read the same SiLU(emb): one GEMM [nb, 4mc] x [4mc, sum(cout)]
  u.g_emb_all = add_gemm(u, emb.w, emb.b, emb.total, ted, 0);
  u.emb_total = emb.total;

  {
    // r5: 34 float slots per transformer block for the static-bound statistics (cs_unet_pack); r6: 4 per AttentionBlock
    int nblk = 0, nab = 0;
    auto count = [&](std::vector<Layer>& layers) {
      for (Layer& l : layers) {
        if (l.kind == ATTN) ++nblk;
        if (l.kind == ATTNBLOCK) ++nab;
      }
    };
    for (auto& layers : u.inp) count(layers);
    count(u.mid);
    for (auto& layers : u.out) count(layers);
    u.extra_slots = (c.math == CS_MATH_F16X3) ? 34 * nblk + 4 * nab : 0;
  }
  layout_arena(u);
  if (u.extra_slots) {
    int k = 0;
    auto assign = [&](std::vector<Layer>& layers) {
      for (Layer& l : layers)
        if (l.kind == ATTN) l.ts_slot = u.extra_slot0 + 34 * k++;
    };
    for (auto& layers : u.inp) assign(layers);
    assign(u.mid);
    for (auto& layers : u.out) assign(layers);
    int k2 = 34 * k;                      // (the AttentionBlocks' slots follow the transformer blocks')
    auto assign2 = [&](std::vector<Layer>& layers) {
      for (Layer& l : layers)
        if (l.kind == ATTNBLOCK) {
          l.ab_slot = u.extra_slot0 + k2;
          k2 += 4;
        }
    };
    for (auto& layers : u.inp) assign2(layers);
    assign2(u.mid);
    for (auto& layers : u.out) assign2(layers);
  }
  return CS_OK;
}

struct Exec : ExecBase {
  const cs_unet& u;
  Exec(const cs_unet& u_, const void* arena_, void* ws_, int64_t ws_bytes_, bool dry_, hipStream_t st_)
      : ExecBase(u_, arena_, ws_, ws_bytes_, dry_, st_), u(u_) {}
  int64_t in_bound_off = -1;      // bound slot of the latent (conv_in's operand), set by forward()

  Act res_block(const Layer& l, const Act& x, const Buf& semb) {
    const int rows = x.d * x.h * x.w;
    // (xb: the skip conv below reads x RAW -- the GroupNorm's finalize kernel leaves x's magnitude bound on the way)
    int64_t xb = -1;
    // (x.d, x.h, x.w: where the conv takes the Winograd-W route the GroupNorm emits that operand, unet.py::_res)
    Buf hn = groupnorm(x.b, l.n[0], x.nb, 1e-5f, CS_ACT_SILU, 32, l.g[0], l.g[2] >= 0 ? &xb : nullptr, x.d, x.h, x.w);
    // (want_stats: the conv's epilogue leaves the partial sums the next GroupNorm takes its statistics from -- unet.py::_res)
    Buf h1 = gemm(hn, l.g[0], x.nb, x.d, x.h, x.w, 1, 0, CS_ACT_NONE, dry ? nullptr : p(semb) + l.emb_lo, semb.c, rows,
                  nullptr, 0, 0, 1, 0, /*want_stats=*/true);
    release(hn);
    Buf hn2 = groupnorm(h1, l.n[1], x.nb, 1e-5f, CS_ACT_SILU, 32, l.g[1], nullptr, x.d, x.h, x.w);
    release(h1);
    Buf skip;
    const bool own_skip = l.g[2] >= 0;
    // (the skip conv reads the RAW residual stream: operand scale from the tensor's actual range, unet.py::_res x_bound=)
    if (own_skip)
      skip = gemm(x.b, l.g[2], x.nb, x.d, x.h, x.w, 1, 0, CS_ACT_NONE, nullptr, 0, 1, nullptr, 0, 0, 1, 0, false, 0.f, xb);
    Act o = x;
    const Buf& sk = own_skip ? skip : x.b;
    o.b = gemm(hn2, l.g[1], x.nb, x.d, x.h, x.w, 1, 0, CS_ACT_NONE, nullptr, 0, 1, dry ? nullptr : p(sk), sk.c, 0, 1, 0,
               /*want_stats=*/true);
    release(hn2);
    if (own_skip) release(skip);
    return o;
  }

  // ResBlock whose input is the concatenation x = [h | skip] with the skip half shared by nb / sk.nb groups of samples
  // (the guidance halves): in_layers' conv is evaluated as a GEMM over channels [0, ks) at the full batch plus a GEMM
  // over channels [ks, C) -- whole GroupNorm groups of skip channels -- at batch sk.nb, which enters the first one's
  // epilogue as a residual (the 1x1x1 skip_connection stays whole).  Same launches, in the same order, as
  // unet.py::_res_split.
  Act res_block_split(const Layer& l, const Act& x, const Act& sk, const Buf& semb) {
    const int rows = x.d * x.h * x.w, C = x.b.c, ks = l.ks, cs = C - ks, off = ks - l.ch_h, cpg = C / 32;
    const int nb = x.nb, nbs = sk.nb, ch_s = sk.b.c, cout = l.cout;
    Act o = x;
    if (nbs <= 0 || nb % nbs || ch_s != C - l.ch_h) {
      chk(CS_EINVAL);
      return o;
    }
    const int64_t m_launch = (int64_t)nbs * rows;
    int64_t xb = -1;
    Buf stats = gn_stats(x.b, nb, 1e-5f, 32, &xb);
    const float* xs = dry ? nullptr : p(sk.b) + off;           // the shared channels of the skip tensor
    // r5: where the halves' convs take the Winograd-W route (each launch covers nbs samples) their operands are emitted in
    // that form -- the h half once per guidance half (unet.py::_res_split)
    const int wn_h = wants_wino(l.gsp[0], nbs, x.d, x.h, x.w), wn_s = wants_wino(l.gsp[1], nbs, x.d, x.h, x.w);
    Buf a_h;
    if (!wn_h)
      a_h = gn_apply_range(dry ? nullptr : p(x.b), C, x.b.rows, nb, stats, l.n[0], 32, cpg, 0, ks, CS_ACT_SILU, l.gsp[0],
                           m_launch);
    Buf a_s = gn_apply_range(xs, ch_s, sk.b.rows, nbs, stats, l.n[0], 32, cpg, ks, cs, CS_ACT_SILU, l.gsp[1], m_launch,
                             wn_s, x.d, x.h, x.w);
    auto lo_of = [&](const Buf& b) -> const void* {            // lo image of a pre-split pair / of a Winograd-W operand
      return ((b.half || b.wino) && !dry) ? reinterpret_cast<const char*>(p(b)) + b.rows * b.c * 2 : nullptr;
    };
    Buf y_s = alloc(sk.b.rows, cout);
    if (ok()) gemm_view(dry ? nullptr : p(a_s), (a_s.half || a_s.wino) ? (dry ? (const void*)1 : lo_of(a_s)) : nullptr, cs,
                        l.gsp[1], nbs, x.d, x.h, x.w, dry ? nullptr : p(y_s), cout, nullptr, 0, 1, nullptr, 0, a_s.a_scale,
                        a_s.wino);
    release(a_s);
    Buf h1 = alloc(x.b.rows, cout);
    for (int g = 0; g < nb / nbs && ok(); ++g) {
      const int64_t r0 = (int64_t)g * nbs * rows;
      const float* rv = dry ? nullptr : p(semb) + (int64_t)g * nbs * semb.c + l.emb_lo;
      if (wn_h) {
        Buf a_g = gn_apply_range(dry ? nullptr : p(x.b) + r0 * C, C, (int64_t)nbs * rows, nbs, stats, l.n[0], 32, cpg, 0, ks,
                                 CS_ACT_SILU, l.gsp[0], m_launch, wn_h, x.d, x.h, x.w, (int64_t)g * nbs);
        if (ok()) gemm_view(dry ? nullptr : p(a_g), dry ? (const void*)1 : lo_of(a_g), ks, l.gsp[0], nbs, x.d, x.h, x.w,
                            dry ? nullptr : p(h1) + r0 * cout, cout, rv, semb.c, rows, dry ? nullptr : p(y_s), cout,
                            a_g.a_scale, a_g.wino);
        release(a_g);
        continue;
      }
      const float* ah = nullptr;
      const void* al = a_h.half ? (const void*)1 : nullptr;
      if (!dry) {
        // fp32: rows of ks floats; pre-split: two fp16 images of ks halves per row
        ah = a_h.half ? reinterpret_cast<const float*>(reinterpret_cast<const char*>(p(a_h)) + r0 * ks * 2) : p(a_h) + r0 * ks;
        if (a_h.half) al = reinterpret_cast<const char*>(lo_of(a_h)) + r0 * ks * 2;
      }
      gemm_view(ah, al, ks, l.gsp[0], nbs, x.d, x.h, x.w, dry ? nullptr : p(h1) + r0 * cout, cout, rv, semb.c, rows,
                dry ? nullptr : p(y_s), cout, a_h.a_scale);
    }
    release(stats);
    if (!wn_h) release(a_h);
    release(y_s);
    Buf skc = gemm(x.b, l.g[2], nb, x.d, x.h, x.w, 1, 0, CS_ACT_NONE, nullptr, 0, 1, nullptr, 0, 0, 1, 0, false, 0.f, xb);
    Buf hn2 = groupnorm(h1, l.n[1], nb, 1e-5f, CS_ACT_SILU, 32, l.g[1], nullptr, x.d, x.h, x.w);
    release(h1);
    o.b = gemm(hn2, l.g[1], nb, x.d, x.h, x.w, 1, 0, CS_ACT_NONE, nullptr, 0, 1, dry ? nullptr : p(skc), cout, 0, 1, 0,
               /*want_stats=*/true);
    release(hn2);
    release(skc);
    return o;
  }

  Act attn_block(const Layer& l, const Act& x, const float* ctxvec) {
    const int c = l.cin, heads = u.cfg.num_heads, dh = c / heads;
    const int n = x.d * x.h * x.w;
    const int64_t rows = (int64_t)x.nb * n;
    Buf xn = groupnorm(x.b, l.n[0], x.nb, 1e-6f, CS_ACT_NONE);
    Buf t0 = linear(xn, l.g[0]);
    release(xn);
    Buf n1 = layernorm(t0, l.n[1]);
    Buf qkv = linear(n1, l.g[1]);
    release(n1);
    Buf a = alloc(rows, c);
    // r5: static bounds of the operands born inside the block (unet.py::_static_scales; the ONE rule in cs_plan.hip)
    float ss[12];
    const Norm& gnn = pl.norms[l.n[0]];
    const bool stat = l.has_ts && u.cfg.math == CS_MATH_F16X3 && !cs_debug()->no_static_scales && ctxvec &&
                      cs_transformer_static_scales(&l.ts, c, n, heads, gnn.gmax, gnn.bmax, l.ctx_max, ss) == CS_OK;
    self_attention(qkv, a, x.nb, n, heads, dh, c, (float)std::pow((double)dh, -0.5), stat ? ss : nullptr);
    if (stat) a.a_scale = ss[3];
    release(qkv);
    // one context token: attn2(x) == to_out(to_v(ctx)) for every query row -> a row vector in this epilogue
    Buf t1 = linear(a, l.g[2], CS_ACT_NONE, ctxvec ? ctxvec + l.ctx_off : nullptr, u.ctx_total, n,
                    dry ? nullptr : p(t0), c);
    release(a);
    release(t0);
    Buf n3 = layernorm(t1, l.n[3]);
    Buf gg;
    // gg's only reader is ff.net.2, t2's only reader proj_out: both producers write the operand pair where they can
    // (unet.py::_attn, out_pair=)
    const float pairs = u.cfg.math == CS_MATH_F16X3 ? 16.f : 0.f;
    const float pair_gg = (pairs > 0.f && stat) ? ss[4] : pairs, pair_t2 = (pairs > 0.f && stat) ? ss[5] : pairs;
    if (l.fused_geglu) {
      gg = linear(n3, l.g[5], CS_ACT_GEGLU, nullptr, 0, 1, nullptr, 0, 0, pair_gg);   // tile 0: a 224-column tile for the gate
    } else {
      Buf ff = linear(n3, l.g[5]);
      gg = alloc(rows, 4 * c);
      if (ok() && !dry) chk(cs_geglu(p(ff), p(gg), (int)rows, 4 * c, 8 * c, 4 * c, st));
      release(ff);
    }
    if (stat && !gg.pair) gg.a_scale = ss[4];      // (a producer that could not emit the pair hands over fp32)
    release(n3);
    Buf t2 = linear(gg, l.g[6], CS_ACT_NONE, nullptr, 0, 1, dry ? nullptr : p(t1), c, 0, pair_t2);
    if (stat && !t2.pair) t2.a_scale = ss[5];
    release(gg);
    release(t1);
    Act o = x;
    // (x.nb samples of n tokens: what the epilogue's GroupNorm partial sums are tiled by -- unet.py::_attn, spatial=)
    o.b = gemm(t2, l.g[7], x.nb, n, 1, 1, 1, 0, CS_ACT_NONE, nullptr, 0, 1, dry ? nullptr : p(x.b), c, 0, 1, 0,
               /*want_stats=*/true);
    release(t2);
    return o;
  }

  Act attnblock(const Layer& l, const Act& x) {
    const int c = l.cin, heads = u.cfg.num_heads, dh = c / heads;
    const int n = x.d * x.h * x.w;
    const int64_t rows = (int64_t)x.nb * n;
    Buf xn = groupnorm(x.b, l.n[0], x.nb, 1e-5f, CS_ACT_NONE);
    Buf qkv = linear(xn, l.g[0]);
    release(xn);
    Buf a = alloc(rows, c);
    // r6: static operand scales of q / k / v and of the attention output (unet.py::_attnblock: the same rule and statistics)
    float ss[4];
    const float qks = (float)std::pow((double)dh, -0.5);
    const bool stat = attnblock_scales(l.n[0], (int64_t)n * (c / 32), c, l.ab_l2max, l.ab_bmax, qks, ss);
    self_attention(qkv, a, x.nb, n, heads, dh, c, qks, stat ? ss : nullptr);
    if (stat) a.a_scale = ss[3];
    release(qkv);
    Act o = x;
    o.b = gemm(a, l.g[1], x.nb, n, 1, 1, 1, 0, CS_ACT_NONE, nullptr, 0, 1, dry ? nullptr : p(x.b), c, 0, 1, 0,
               /*want_stats=*/true);
    release(a);
    return o;
  }

  // runs the layers of one block; `keep_in` says whether the caller still needs the input buffer
  Act run(const std::vector<Layer>& layers, Act h, const Buf& semb, const float* ctxvec, bool keep_in,
          const Act* split_skip = nullptr) {
    bool owned = !keep_in;
    for (const Layer& l : layers) {
      Act o;
      if (l.kind == RES && l.ks > 0 && split_skip && &l == &layers[0]) {
        o = res_block_split(l, h, *split_skip, semb);
        if (owned) release(h.b);
        h = o;
        owned = true;
        continue;
      }
      switch (l.kind) {
        case CONV_IN:
          o = h;
          o.b = gemm(h.b, l.g[0], h.nb, h.d, h.h, h.w, 1, 0, CS_ACT_NONE, nullptr, 0, 1, nullptr, 0, 0, 1, 0, true, 0.f,
                     in_bound_off);
          break;
        case RES:
          o = res_block(l, h, semb);
          break;
        case ATTN:
          o = attn_block(l, h, ctxvec);
          break;
        case ATTNBLOCK:
          o = attnblock(l, h);
          break;
        case DOWN: {   // dims == 3: inner two dims only (openai_model_3d.py:188); dims == 4: all three
          const int sd = u.cfg.dims == 3 ? 1 : 2;
          o = h;
          // (Down / Upsample read the RAW stream with no GroupNorm in front: its bound from the producers' partials)
          o.b = gemm(h.b, l.g[0], h.nb, h.d, h.h, h.w, 2, 0, CS_ACT_NONE, nullptr, 0, 1, nullptr, 0, 0, sd, 0, true, 0.f,
                     range_bound(h.b, h.nb));
          o.d = h.d / sd;
          o.h = h.h / 2;
          o.w = h.w / 2;
          break;
        }
        case UP: {     // nearest x2 folded into the conv's addressing (openai_model_3d.py:148-157)
          const int ud = u.cfg.dims == 3 ? 0 : 1;
          o = h;
          o.b = gemm(h.b, l.g[0], h.nb, h.d, h.h, h.w, 1, 1, CS_ACT_NONE, nullptr, 0, 1, nullptr, 0, 0, 1, ud, true, 0.f,
                     range_bound(h.b, h.nb));
          o.d = h.d << ud;
          o.h = h.h * 2;
          o.w = h.w * 2;
          break;
        }
      }
      if (owned) release(h.b);
      h = o;
      owned = true;
    }
    return h;
  }

  Buf duplicate(const Buf& a) {   // torch.cat([a, a], dim=0); the copy gets its own copy of the producer's partials
    Buf o = alloc(2 * a.rows, a.c);
    if (a.nseg == 1 && a.seg[0].valid()) {
      o.seg[0] = dup_stat(a.seg[0]);
      o.nseg = o.seg[0].valid() ? 1 : 0;
    }
    if (ok() && !dry) {
      const size_t bytes = (size_t)a.rows * a.c * 4;
      if (hipMemcpyAsync(p(o), p(a), bytes, hipMemcpyDeviceToDevice, st) != hipSuccess ||
          hipMemcpyAsync((char*)p(o) + bytes, p(a), bytes, hipMemcpyDeviceToDevice, st) != hipSuccess)
        chk(CS_EINVAL);
    }
    return o;
  }
};

bool has_attn(const std::vector<Layer>& layers) {
  for (const Layer& l : layers)
    if (l.kind == ATTN || l.kind == ATTNBLOCK) return true;
  return false;
}

int forward(Exec& e, const float* x_ncdhw, const int64_t* t, const float* ctxvec, float* eps_ncdhw, int nbx,
            int cfg_pairs) {
  const cs_unet& u = e.u;
  const CsUnetConfig& c = u.cfg;
  const int S = c.d * c.h * c.w;
  e.sync_begin();         // split-K arrival counters (CsConvGemm.splitk_sync)
  e.amax_begin(64);       // magnitude-bound slots of this forward (unet.py::forward_ndhwc: self._amax)
  Buf temb = e.alloc(nbx, c.model_channels);
  if (e.ok() && !e.dry) e.chk(cs_timestep_embedding(t, e.p(temb), nbx, c.model_channels, 10000.0f, e.st));
  Buf e1 = e.linear(temb, u.g_te0, CS_ACT_SILU);
  e.release(temb);
  // every consumer of `emb` is emb_layers = SiLU -> Linear (openai_model_3d.py:257-263): keep SiLU(emb)
  Buf e2 = e.linear(e1, u.g_te2, CS_ACT_SILU);
  e.release(e1);
  Buf semb = e.linear(e2, u.g_emb_all);
  e.release(e2);
  Act h;
  h.nb = nbx; h.d = c.d; h.h = c.h; h.w = c.w;
  h.b = e.alloc((int64_t)nbx * S, u.cpad_in);
  if (e.ok() && !e.dry) e.chk(cs_nchw_to_ndhwc(x_ncdhw, e.p(h.b), nbx, c.in_channels, S, u.cpad_in, e.st));
  // r6: conv_in reads the RAW latent -- its exact max |.| goes to a bound slot (unet.py::forward_ndhwc: ops.absmax_bound)
  e.in_bound_off = e.amax_slot();
  if (e.in_bound_off >= 0 && e.ok() && !e.dry)
    e.chk(cs_absmax(e.p(h.b), (int64_t)nbx * S * u.cpad_in, e.bound_ptr(e.in_bound_off), e.st));
  bool shared = cfg_pairs != 0;
  auto split = [&](bool h_retained) {   // first context-dependent block: one copy per guidance half, [uc; c]
    Buf h2 = e.duplicate(h.b);
    if (!h_retained) e.release(h.b);     // otherwise it lives on in `hs` as a (shared) skip tensor
    h.b = h2;
    h.nb *= 2;
    Buf s2 = e.duplicate(semb);
    e.release(semb);
    semb = s2;
    shared = false;
  };
  std::vector<Act> hs;
  bool first = true;
  for (const auto& layers : u.inp) {
    bool keep = !first;           // the block input is the previous block's output, retained in `hs`
    if (shared && has_attn(layers)) {
      split(!first);              // the duplicate is a fresh buffer this block may consume
      keep = false;
    }
    h = e.run(layers, h, semb, ctxvec, keep);
    hs.push_back(h);
    first = false;
    if (!e.ok()) return e.rc;
  }
  bool keep_mid = true;           // h is hs.back()
  if (shared) {
    split(true);
    keep_mid = false;
  }
  h = e.run(u.mid, h, semb, ctxvec, keep_mid);
  for (const auto& layers : u.out) {
    Act sk = hs.back();
    hs.pop_back();
    // torch.cat([h, skip], channel); a shared (nbx-sized) skip tensor feeds both guidance halves
    Act cat = h;
    cat.b = e.alloc(h.b.rows, h.b.c + sk.b.c);
    if (e.ok() && !e.dry) {
      e.chk(cs_copy_rows(e.p(h.b), e.p(cat.b), h.b.rows, h.b.c, h.b.c, cat.b.c, e.st));
      const int groups = h.nb / sk.nb;
      if (groups * sk.nb != h.nb) e.chk(CS_EINVAL);
      for (int g = 0; g < groups && e.ok(); ++g)
        e.chk(cs_copy_rows(e.p(sk.b), e.p(cat.b) + (int64_t)g * sk.b.rows * cat.b.c + h.b.c, sk.b.rows, sk.b.c,
                           sk.b.c, cat.b.c, e.st));
    }
    // the concatenation's GroupNorm takes its statistics from both halves' producers (unet.py: cs_segs): their partials
    // move to the concatenation buffer and are released with it
    cat.b.nseg = 0;
    if (h.b.nseg == 1 && sk.b.nseg == 1 && h.b.seg[0].valid() && sk.b.seg[0].valid()) {
      cat.b.seg[0] = h.b.seg[0];
      cat.b.seg[1] = sk.b.seg[0];
      cat.b.nseg = 2;
      h.b.seg[0] = Stat();
      sk.b.seg[0] = Stat();
      h.b.nseg = sk.b.nseg = 0;
    }
    e.release(h.b);
    if (layers[0].kind == RES && layers[0].ks > 0 && cat.b.rows >= u.split_min_rows) {
      h = e.run(layers, cat, semb, ctxvec, false, &sk);      // the skip tensor itself feeds the shared GEMMs
      e.release(sk.b);
    } else {
      e.release(sk.b);
      h = e.run(layers, cat, semb, ctxvec, false);
    }
    if (!e.ok()) return e.rc;
  }
  Buf hn = e.groupnorm(h.b, u.n_out, h.nb, 1e-5f, CS_ACT_SILU, 32, u.g_out);
  e.release(h.b);
  Buf eps = e.gemm(hn, u.g_out, h.nb, h.d, h.h, h.w);
  e.release(hn);
  if (e.ok() && !e.dry) e.chk(cs_ndhwc_to_nchw(e.p(eps), eps_ncdhw, h.nb, c.out_channels, S, eps.c, e.st));
  e.release(eps);
  e.release(semb);
  e.release(e.amax_arena);
  return e.rc;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------
extern "C" int cs_unet_create(const CsUnetConfig* cfg, cs_unet** out) {
  if (!cfg || !out) return CS_EINVAL;
  cs_unet* u = new (std::nothrow) cs_unet();
  if (!u) return CS_ENOMEM;
  u->cfg = *cfg;
  u->math = cfg->math;
  const int rc = build(*u);
  if (rc != CS_OK) {
    delete u;
    return rc;
  }
  *out = u;
  return CS_OK;
}

extern "C" void cs_unet_destroy(cs_unet* u) { delete u; }

extern "C" int cs_unet_param_count(const cs_unet* u) { return u ? (int)u->params.size() : 0; }

extern "C" int cs_unet_param_info(const cs_unet* u, int i, const char** name, int64_t shape5[5], int* ndim,
                                  int64_t* raw_offset_bytes) {
  return plan_param_info(u, i, name, shape5, ndim, raw_offset_bytes);
}

extern "C" int64_t cs_unet_raw_bytes(const cs_unet* u) { return u ? u->raw_bytes : 0; }
extern "C" int64_t cs_unet_arena_bytes(const cs_unet* u) { return u ? u->arena_bytes : 0; }
extern "C" int64_t cs_unet_context_floats(const cs_unet* u) { return u ? u->ctx_total : 0; }

extern "C" int cs_unet_pack(cs_unet* u, const void* raw_dev, void* arena_dev, cs_stream_t stream) {
  const int rc = pack_plan(u, raw_dev, arena_dev, stream);
  if (rc != CS_OK || !u || u->math != CS_MATH_F16X3) return rc;
  {
    // r6: every AttentionBlock's static-bound statistics (unet.py::_pack: the same kernel on the same tensors, the same values):
    // {max row 2-norm, max |.|} of the fused qkv weight into slots 0-1, of its bias into slots 2-3; one read-back
    hipStream_t st0 = (hipStream_t)stream;
    const char* raw0 = reinterpret_cast<const char*>(raw_dev);
    float* d0 = reinterpret_cast<float*>(reinterpret_cast<char*>(arena_dev) + u->amax_off);
    std::vector<Layer*> abs_;
    auto collect_ab = [&](std::vector<Layer>& layers) {
      for (Layer& l : layers)
        if (l.kind == ATTNBLOCK && l.ab_slot >= 0) abs_.push_back(&l);
    };
    for (auto& layers : u->inp) collect_ab(layers);
    collect_ab(u->mid);
    for (auto& layers : u->out) collect_ab(layers);
    for (Layer* lp : abs_) {
      const int c3 = 3 * lp->cin;
      int r2 = cs_weight_rowstats(reinterpret_cast<const float*>(raw0 + u->params[lp->ab_w].raw_off), c3, lp->cin, d0 + lp->ab_slot,
                                  stream);
      if (r2 == CS_OK)
        r2 = cs_weight_rowstats(reinterpret_cast<const float*>(raw0 + u->params[lp->ab_b].raw_off), 1, c3, d0 + lp->ab_slot + 2,
                                stream);
      if (r2 != CS_OK) return r2;
    }
    if (!abs_.empty()) {
      std::vector<float> host0(4 * abs_.size());
      if (hipMemcpyAsync(host0.data(), d0 + abs_[0]->ab_slot, host0.size() * 4, hipMemcpyDeviceToHost, st0) != hipSuccess ||
          hipStreamSynchronize(st0) != hipSuccess)
        return CS_EINVAL;
      for (size_t i = 0; i < abs_.size(); ++i) {
        abs_[i]->ab_l2max = host0[4 * i];
        abs_[i]->ab_bmax = host0[4 * i + 3];
      }
    }
  }
  if (!u->cfg.use_spatial_transformer) return rc;
  // r5: the static-bound statistics of every transformer block (unet.py::_pack: the same kernel, the same values) -- row
  // 2-norm maxima and |.| maxima of the block's Linears and LayerNorm parameters; 34 float slots per block behind the
  // plan's |w| maxima (zeroed by pack_plan), ONE more read-back at load time
  hipStream_t st = (hipStream_t)stream;
  const char* raw = reinterpret_cast<const char*>(raw_dev);
  float* d_amax = reinterpret_cast<float*>(reinterpret_cast<char*>(arena_dev) + u->amax_off);
  auto src = [&](int param) { return reinterpret_cast<const float*>(raw + u->params[param].raw_off); };
  std::vector<Layer*> blocks;
  auto collect = [&](std::vector<Layer>& layers) {
    for (Layer& l : layers)
      if (l.kind == ATTN && l.ts_slot >= 0) blocks.push_back(&l);
  };
  for (auto& layers : u->inp) collect(layers);
  collect(u->mid);
  for (auto& layers : u->out) collect(layers);
  if (blocks.empty()) return rc;
  for (Layer* lp : blocks) {
    const Layer& l = *lp;
    const int c = l.cin;
    float* o = d_amax + l.ts_slot;
    const Gemm& qkv = u->gemms[l.g[1]];
    const Gemm& to = u->gemms[l.g[2]];
    const Gemm& ff = u->gemms[l.g[5]];
    const Gemm& f2 = u->gemms[l.g[6]];
    const Gemm& pi = u->gemms[l.g[0]];
    const Norm& n1 = u->norms[l.n[1]];
    const Norm& n3 = u->norms[l.n[3]];
    if (qkv.w.size() != 3 || to.w.empty() || to.b.empty() || ff.w.empty() || ff.b.empty() || f2.w.empty() || f2.b.empty() ||
        pi.w.empty() || pi.b.empty())
      return CS_EINVAL;
    const float* wff = src(ff.w[0].param);
    const float* bff = src(ff.b[0].param);
    struct Job { const float* w; int rows, cols; };
    const Job jobs[17] = {
        {src(qkv.w[0].param), c, c}, {src(qkv.w[1].param), c, c}, {src(qkv.w[2].param), c, c},        // rq rk rv
        {src(to.w[0].param), c, c}, {src(to.b[0].param), 1, c},                                          // ro bo
        {wff, 4 * c, c}, {bff, 1, 4 * c}, {wff + (int64_t)4 * c * c, 4 * c, c}, {bff + 4 * c, 1, 4 * c},  // rx bx rg bg
        {src(f2.w[0].param), c, 4 * c}, {src(f2.b[0].param), 1, c},                                      // r2 b2
        {src(pi.w[0].param), c, c}, {src(pi.b[0].param), 1, c},                                          // rpi bpi
        {src(n1.gp), 1, c}, {src(n1.bp), 1, c}, {src(n3.gp), 1, c}, {src(n3.bp), 1, c}};                 // g1 be1 g3 be3
    for (int j = 0; j < 17; ++j) {
      const int r2 = cs_weight_rowstats(jobs[j].w, jobs[j].rows, jobs[j].cols, o + 2 * j, stream);
      if (r2 != CS_OK) return r2;
    }
  }
  std::vector<float> host((size_t)34 * blocks.size());
  // (the blocks' slots are contiguous: they were reserved one after the other)
  if (hipMemcpyAsync(host.data(), d_amax + blocks[0]->ts_slot, host.size() * 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess)
    return CS_EINVAL;
  for (size_t b = 0; b < blocks.size(); ++b) {
    const float* h = host.data() + 34 * b;
    // which of {row norm, abs max} each field takes (unet.py::_pack)
    static const int which[17] = {0, 0, 0, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 1, 0, 1, 0};
    float v[17];
    for (int j = 0; j < 17; ++j) v[j] = h[2 * j + which[j]];
    CsTransformerStats& t = blocks[b]->ts;
    t.rq = v[0]; t.rk = v[1]; t.rv = v[2]; t.ro = v[3]; t.bo = v[4]; t.rx = v[5]; t.bx = v[6]; t.rg = v[7]; t.bg = v[8];
    t.r2 = v[9]; t.b2 = v[10]; t.rpi = v[11]; t.bpi = v[12]; t.g1 = v[13]; t.be1 = v[14]; t.g3 = v[15]; t.be3 = v[16];
    blocks[b]->has_ts = true;
  }
  return CS_OK;
}

// r5: the largest |entry| of every transformer block's cross-attention row vector of the CURRENT run, in block order (input
// blocks, middle, output blocks) -- host data the caller reads back once per run after cs_unet_context (it enters t1's
// static bound, cs_transformer_static_scales).  n must equal the number of transformer blocks; never called: 0 is assumed.
extern "C" int cs_unet_set_context_bounds(cs_unet* u, const float* ctx_max, int n) {
  if (!u || !ctx_max || n < 0) return CS_EINVAL;
  std::vector<Layer*> blocks;
  auto collect = [&](std::vector<Layer>& layers) {
    for (Layer& l : layers)
      if (l.kind == ATTN) blocks.push_back(&l);
  };
  for (auto& layers : u->inp) collect(layers);
  collect(u->mid);
  for (auto& layers : u->out) collect(layers);
  if ((int)blocks.size() != n) return CS_EINVAL;
  for (int i = 0; i < n; ++i) blocks[i]->ctx_max = ctx_max[i] > 0.f ? ctx_max[i] : 0.f;
  return CS_OK;
}

extern "C" int64_t cs_unet_workspace_bytes(const cs_unet* u, int nb_x, int cfg_pairs) {
  if (!u || nb_x <= 0) return CS_EINVAL;
  Exec e(*u, nullptr, nullptr, 0, true, nullptr);
  const int rc = forward(e, nullptr, nullptr, nullptr, nullptr, nb_x, cfg_pairs);
  if (rc != CS_OK) return rc;
  // the context pass needs one [nb_ctx][max c] temporary
  int maxc = 0;
  for (const Gemm& g : u->gemms) maxc = g.cout > maxc && g.cin == u->cfg.context_dim ? g.cout : maxc;
  const int64_t ctx_ws = align_up((int64_t)(cfg_pairs ? 2 : 1) * nb_x * maxc * 4) + 3 * ALIGN;      // (+ the bound slots)
  return e.peak > ctx_ws ? e.peak : ctx_ws;
}

extern "C" int cs_unet_context(const cs_unet* u, const void* arena, const float* ctx, int nb_ctx, float* ctxvec,
                               int32_t* status, void* workspace, int64_t workspace_bytes, cs_stream_t stream) {
  if (!u || !u->packed || !arena || !ctx || !ctxvec || !workspace || nb_ctx <= 0) return CS_EINVAL;
  if (!u->cfg.use_spatial_transformer) return CS_EINVAL;      // the concat family has no context
  Exec e(*u, arena, workspace, workspace_bytes, false, (hipStream_t)stream);
  const bool dyn = u->cfg.math == CS_MATH_F16X3 && !cs_debug()->no_dyn_scale && !cs_debug()->no_static_scales;
  Buf slots = e.alloc(64, 1);              // slot 0: max |ctx|; slot 1 + k: max |to_v(ctx)| of transformer block k
  int nslot = 0;
  if (dyn && e.ok() && hipMemsetAsync(e.p(slots), 0, 64 * 4, e.st) != hipSuccess) e.chk(CS_EINVAL);
  // ctx rows are read in place: describe them as a buffer view at offset (ctx - workspace)
  auto visit = [&](const Layer& l) {
    if (l.kind != ATTN || !e.ok()) return;
    const Gemm& gv = u->gemms[l.g[3]];
    const Gemm& go = u->gemms[l.g[4]];
    Buf v2 = e.alloc(nb_ctx, gv.cout);
    if (!e.ok()) return;
    CsConvGemm q;
    for (int pass = 0; pass < 2 && e.ok(); ++pass) {
      const Gemm& g = pass == 0 ? gv : go;
      memset(&q, 0, sizeof(q));
      // r5 (unet.py::_context_vectors): the context and to_v's output are raw operands -- their scale follows a device-side
      // magnitude bound (max |.| into a workspace slot, CsConvGemm.a_bound) instead of the constant 16
      if (dyn) {
        const float* src = pass == 0 ? ctx : e.p(v2);
        const int64_t nel = (int64_t)nb_ctx * (pass == 0 ? u->cfg.context_dim : gv.cout);
        float* slot = e.p(slots) + (pass == 0 ? 0 : 1 + nslot);
        if (pass == 1 || nslot == 0) {
          CS_LAUNCH(absmax_kernel, dim3(cs_grid_for(nel, 256, 256)), dim3(256), 0, e.st, src, nel, slot);
          if (hipGetLastError() != hipSuccess) e.chk(CS_EINVAL);
        }
        q.a_bound = slot;
        if (pass == 1) ++nslot;
      }
      q.x = pass == 0 ? ctx : e.p(v2);
      q.out = pass == 0 ? e.p(v2) : ctxvec + l.ctx_off;
      q.w = reinterpret_cast<const float*>(e.arena + g.w_off);
      if (u->cfg.math == CS_MATH_F16X3) {
        q.w_lo = e.arena + g.wlo_off;
        q.acc_scale = g.acc_scale;
        q.a_scale = 16.0f;
      }
      q.bias = g.b_off >= 0 ? e.wf(g.b_off) : nullptr;
      q.nb = nb_ctx; q.din = q.hin = q.win = q.dout = q.hout = q.wout = 1;
      q.cin = g.cin_pad; q.cout = g.cout;
      q.lda = g.cin_pad; q.ldw = g.ldw; q.ldo = pass == 0 ? g.cout : u->ctx_total;
      q.kd = q.kh = q.kw = 1; q.sd = q.sh = q.sw = 1;
      q.rv_rows = 1; q.math = u->cfg.math;
      q.status = status;
      e.chk(cs_conv_gemm(&q, e.st));
    }
    e.release(v2);
  };
  for (const auto& layers : u->inp)
    for (const Layer& l : layers) visit(l);
  for (const Layer& l : u->mid) visit(l);
  for (const auto& layers : u->out)
    for (const Layer& l : layers) visit(l);
  return e.rc;
}

extern "C" int cs_unet_step(const cs_unet* u, const void* arena, const float* x_ncdhw, const int64_t* t,
                            const float* ctxvec, float* eps_ncdhw, int nb_x, int cfg_pairs, int32_t* status,
                            void* workspace, int64_t workspace_bytes, cs_stream_t stream) {
  if (!u || !u->packed || !arena || !x_ncdhw || !t || !eps_ncdhw || !workspace || nb_x <= 0) return CS_EINVAL;
  // crossattn family: ctxvec (cs_unet_context) is required.  concat family (no transformer blocks): there is no
  // context, x carries the condition volume as its last channel(s) (network.py:25-27) and the guidance halves
  // share nothing, so cfg_pairs must be 0 and the caller passes the duplicated batch.
  if (u->cfg.use_spatial_transformer ? !ctxvec : (ctxvec != nullptr || cfg_pairs != 0)) return CS_EINVAL;
  if (((uintptr_t)workspace & 15) || ((uintptr_t)arena & 15) || ((uintptr_t)ctxvec & 15)) return CS_EINVAL;
  Exec e(*u, arena, workspace, workspace_bytes, false, (hipStream_t)stream);
  e.status = status;
  return forward(e, x_ncdhw, t, ctxvec, eps_ncdhw, nb_x, cfg_pairs);
}
