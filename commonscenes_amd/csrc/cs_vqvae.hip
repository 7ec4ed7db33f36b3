// cs_vqvae_*: VQVAE.decode_no_quant / decode as ONE native call over a packed weight arena -- the downstream end
// of the hot path, sequenced from host C++ exactly as commonscenes_amd/vqvae.py does, so the
// two drivers agree bit for bit.
//
// Reference being replaced:
//   model/networks/vqvae_networks/network.py:90-103        VQVAE.decode / decode_no_quant
//   model/networks/vqvae_networks/quantizer.py:68-119      VectorQuantizer.forward (is_voxel=True): nearest code
//   model/networks/vqvae_networks/vqvae_modules.py:292-409 Decoder3D (+ ResnetBlock :64-123, AttnBlock :128-178,
//                                                          Upsample :24-39, Normalize :13-21)
// Scope: config/vqvae_snet.yaml (attn_resolutions empty: attention in the mid block only).
#include "cs_driver.h"

namespace {

int vq_groups(int c) {   // vqvae_modules.py:13-21 (Normalize): 32 groups, fewer for narrow tensors
  if (c <= 32) return c / 4;
  if (c % 32 != 0) return 30;
  return 32;
}

struct ResP {
  int n1, n2, c1, c2, nin;   // norms, convs, 1x1 shortcut (-1: identity)
  int cin, cout;
};

}  // namespace

struct cs_vqvae : Plan {
  CsVqvaeConfig cfg;
  int g_conv_in = -1, g_qkv = -1, g_proj = -1, g_conv_out = -1, g_post = -1;
  int n_attn = -1, n_out = -1, c_book = -1;
  ResP mid1, mid2;
  std::vector<std::vector<ResP>> up;      // [level][block]
  std::vector<int> up_conv;               // [level] upsample conv GEMM or -1
  int block_in0 = 0, c_final = 0, grid = 0;
  // r6: what bounds mid.attn_1's q / k / v (cs_attnblock_static_scales): max row 2-norm of the three 1x1x1 conv weights and
  // max |bias|, filled by cs_vqvae_pack from the raw weights (F16X3 only; 0 = none: the constant scale 16)
  int p_qkv_w[3] = {-1, -1, -1}, p_qkv_b[3] = {-1, -1, -1};
  float attn_l2max = 0.f, attn_bmax = 0.f;
};

namespace {

ResP make_res(cs_vqvae& u, const std::string& p, int cin, int cout) {
  ResP r;
  r.cin = cin;
  r.cout = cout;
  r.n1 = add_norm(u, p + ".norm1", cin);
  r.c1 = add_layer_gemm(u, p + ".conv1", cout, cin, 3);
  r.n2 = add_norm(u, p + ".norm2", cout);
  r.c2 = add_layer_gemm(u, p + ".conv2", cout, cout, 3);
  r.nin = cin != cout ? add_layer_gemm(u, p + ".nin_shortcut", cout, cin, 1) : -1;
  return r;
}

int build(cs_vqvae& u) {
  const CsVqvaeConfig& c = u.cfg;
  if (c.ch <= 0 || c.out_ch <= 0 || c.n_mult <= 0 || c.n_mult > 8 || c.num_res_blocks <= 0 || c.z_channels <= 0 ||
      c.resolution <= 0 || c.n_embed <= 0 || c.embed_dim <= 0 || c.embed_dim > 3)
    return CS_EINVAL;
  if (c.math != CS_MATH_FP32 && c.math != CS_MATH_F16X3) return CS_EINVAL;
  if ((c.resolution >> (c.n_mult - 1)) << (c.n_mult - 1) != c.resolution) return CS_EINVAL;
  for (int i = 0; i < c.n_mult; ++i)
    if (c.ch_mult[i] <= 0 || (c.ch * c.ch_mult[i]) % 4) return CS_EINVAL;
  u.grid = c.resolution >> (c.n_mult - 1);
  const std::string D = "decoder.";
  int block_in = c.ch * c.ch_mult[c.n_mult - 1];
  u.block_in0 = block_in;
  u.g_conv_in = add_layer_gemm(u, D + "conv_in", block_in, c.z_channels, 3);
  u.mid1 = make_res(u, D + "mid.block_1", block_in, block_in);
  u.n_attn = add_norm(u, D + "mid.attn_1.norm", block_in);
  int w[3], b[3];
  const char* qkv[3] = {"q", "k", "v"};
  for (int j = 0; j < 3; ++j) add_wb(u, D + "mid.attn_1." + qkv[j], block_in, block_in, 1, true, w[j], b[j]);
  u.g_qkv = add_gemm(u, {{w[0], 0, block_in}, {w[1], 0, block_in}, {w[2], 0, block_in}},
                     {{b[0], 0, block_in}, {b[1], 0, block_in}, {b[2], 0, block_in}}, 3 * block_in, block_in, 1);
  for (int j = 0; j < 3; ++j) {
    u.p_qkv_w[j] = w[j];
    u.p_qkv_b[j] = b[j];
  }
  u.g_proj = add_layer_gemm(u, D + "mid.attn_1.proj_out", block_in, block_in, 1);
  u.mid2 = make_res(u, D + "mid.block_2", block_in, block_in);
  u.up.assign(c.n_mult, {});
  u.up_conv.assign(c.n_mult, -1);
  for (int lvl = c.n_mult - 1; lvl >= 0; --lvl) {
    const int block_out = c.ch * c.ch_mult[lvl];
    for (int blk = 0; blk < c.num_res_blocks; ++blk) {
      u.up[lvl].push_back(make_res(u, D + "up." + std::to_string(lvl) + ".block." + std::to_string(blk), block_in,
                                   block_out));
      block_in = block_out;
    }
    if (lvl != 0)
      u.up_conv[lvl] = add_layer_gemm(u, D + "up." + std::to_string(lvl) + ".upsample.conv", block_in, block_in, 3,
                                      true, 0, 7);
  }
  u.c_final = block_in;
  u.n_out = add_norm(u, D + "norm_out", block_in);
  u.g_conv_out = add_layer_gemm(u, D + "conv_out", c.out_ch, block_in, 3, true, 0, 0, /*tapcol=*/true);
  const int book = add_param(u, "quantize.embedding.weight", {c.n_embed, c.embed_dim});
  u.c_book = add_copy(u, book);
  // post_quant_conv (network.py:92): emits a zero 4th channel so conv_in reads float4-aligned rows
  int wp, bp;
  add_wb(u, "post_quant_conv", c.z_channels, c.embed_dim, 1, true, wp, bp);
  const int zpad = (4 - c.z_channels % 4) % 4;
  std::vector<Piece> pw = {{wp, 0, c.z_channels}}, pb = {{bp, 0, c.z_channels}};
  if (zpad) {
    pw.push_back({-1, 0, zpad});
    pb.push_back({-1, 0, zpad});
  }
  u.g_post = add_gemm(u, pw, pb, c.z_channels + zpad, c.embed_dim, 1);
  u.extra_slots = c.math == CS_MATH_F16X3 ? 4 : 0;        // r6: the attention block's static-bound statistics (cs_vqvae_pack)
  layout_arena(u);
  return CS_OK;
}

struct VExec : ExecBase {
  const cs_vqvae& u;
  VExec(const cs_vqvae& u_, const void* arena_, void* ws_, int64_t ws_bytes_, bool dry_, hipStream_t st_)
      : ExecBase(u_, arena_, ws_, ws_bytes_, dry_, st_), u(u_) {
    stats_invariant_only = true;      // r5: GroupNorm partials only from batch-independent statistics tiles (vqvae.py)
  }

  // ResnetBlock.forward (vqvae_modules.py:103-123): GN+swish -> conv -> GN+swish -> conv, + (1x1-projected) input
  Act res(const ResP& r, const Act& x) {
    // (x.d, x.h, x.w, r5: the Winograd-W operand where the conv takes that route -- decided by the sample geometry, vqvae.py)
    Buf h = groupnorm(x.b, r.n1, x.nb, 1e-6f, CS_ACT_SILU, vq_groups(r.cin), r.c1, nullptr, x.d, x.h, x.w);
    // (want_stats: the conv's epilogue leaves the partial sums the next GroupNorm takes its statistics from -- vqvae.py::_res)
    Buf h1 = gemm(h, r.c1, x.nb, x.d, x.h, x.w, 1, 0, CS_ACT_NONE, nullptr, 0, 1, nullptr, 0, 0, 1, 0, /*want_stats=*/true);
    release(h);
    Buf h2 = groupnorm(h1, r.n2, x.nb, 1e-6f, CS_ACT_SILU, vq_groups(r.cout), r.c2, nullptr, x.d, x.h, x.w);
    release(h1);
    Buf skip = x.b;
    if (r.nin >= 0) skip = gemm(x.b, r.nin, x.nb, x.d, x.h, x.w);
    Act o = x;
    o.b = gemm(h2, r.c2, x.nb, x.d, x.h, x.w, 1, 0, CS_ACT_NONE, nullptr, 0, 1, dry ? nullptr : p(skip), skip.c, 0, 1, 0,
               /*want_stats=*/true);
    release(h2);
    if (r.nin >= 0) release(skip);
    return o;
  }

  // AttnBlock.forward (vqvae_modules.py:154-178): single head over all voxels
  Act attn(const Act& x) {
    const int c = x.b.c, n = x.d * x.h * x.w;
    const int64_t rows = (int64_t)x.nb * n;
    Buf hn = groupnorm(x.b, u.n_attn, x.nb, 1e-6f, CS_ACT_NONE, vq_groups(c));
    Buf qkv = linear(hn, u.g_qkv);
    release(hn);
    Buf a = alloc(rows, c);
    // r6: static operand scales of q / k / v and of the attention output (vqvae.py::_attn: the same rule, the same statistics)
    float ss[4];
    const float qks = (float)std::pow((double)c, -0.5);
    const bool stat = attnblock_scales(u.n_attn, (int64_t)n * (c / vq_groups(c)), c, u.attn_l2max, u.attn_bmax, qks, ss);
    self_attention(qkv, a, x.nb, n, 1, c, c, qks, stat ? ss : nullptr);
    if (stat) a.a_scale = ss[3];
    release(qkv);
    Act o = x;
    o.b = linear(a, u.g_proj, CS_ACT_NONE, nullptr, 0, 1, dry ? nullptr : p(x.b), c);
    release(a);
    return o;
  }
};

int decode(VExec& e, const float* latent_ncdhw, float* sdf_ncdhw, int64_t* idx_out, int nb, int quantize) {
  const cs_vqvae& u = e.u;
  const CsVqvaeConfig& c = u.cfg;
  const int g = u.grid, S = g * g * g;
  const int64_t rows = (int64_t)nb * S;
  Buf zl = e.alloc(rows, 4);
  if (e.ok() && !e.dry) e.chk(cs_nchw_to_ndhwc(latent_ncdhw, e.p(zl), nb, c.embed_dim, S, 4, e.st));
  if (quantize) {   // quantizer.py:76-84: nearest codebook row; the 4th (padding) channel stays zero
    Buf zq = e.alloc(rows, 4);
    Buf idx = idx_out ? Buf() : e.alloc(rows * 2, 1);      // int64 scratch when the caller does not want the indices
    if (e.ok() && !e.dry) {
      if (hipMemsetAsync(e.p(zq), 0, (size_t)rows * 16, e.st) != hipSuccess) e.chk(CS_EINVAL);
      int64_t* ip = idx_out ? idx_out : reinterpret_cast<int64_t*>(e.p(idx));
      e.chk(cs_vq_argmin_lookup(e.p(zl), e.wf(u.copies[u.c_book].arena_off), ip, e.p(zq), rows, c.n_embed, c.embed_dim,
                                4, 4, e.st));
    }
    if (!idx_out) e.release(idx);
    e.release(zl);
    zl = zq;
  }
  Act h;
  h.nb = nb; h.d = h.h = h.w = g;
  Buf q4 = e.gemm(zl, u.g_post, nb, g, g, g);              // post_quant_conv (1x1x1)
  e.release(zl);
  h.b = e.gemm(q4, u.g_conv_in, nb, g, g, g, 1, 0, CS_ACT_NONE, nullptr, 0, 1, nullptr, 0, 0, 1, 0, /*want_stats=*/true);
  e.release(q4);
  auto step = [&](Act o) {
    e.release(h.b);
    h = o;
  };
  step(e.res(u.mid1, h));
  step(e.attn(h));
  step(e.res(u.mid2, h));
  for (int lvl = c.n_mult - 1; lvl >= 0 && e.ok(); --lvl) {
    for (const ResP& r : u.up[lvl]) step(e.res(r, h));
    if (u.up_conv[lvl] >= 0) {     // Upsample (vqvae_modules.py:35-39): nearest x2 in D, H, W as conv addressing
      Act o = h;
      o.b = e.gemm(h.b, u.up_conv[lvl], h.nb, h.d, h.h, h.w, 1, 1, CS_ACT_NONE, nullptr, 0, 1, nullptr, 0, 0, 1, 1);
      o.d = h.d * 2; o.h = h.h * 2; o.w = h.w * 2;
      step(o);
    }
  }
  Buf hn = e.groupnorm(h.b, u.n_out, h.nb, 1e-6f, CS_ACT_GELU, vq_groups(u.c_final), u.g_conv_out);
  e.release(h.b);
  Buf out = e.gemm(hn, u.g_conv_out, h.nb, h.d, h.h, h.w);
  e.release(hn);
  if (e.ok() && !e.dry)
    e.chk(cs_ndhwc_to_nchw(e.p(out), sdf_ncdhw, h.nb, c.out_ch, h.d * h.h * h.w, out.c, e.st));
  e.release(out);
  return e.rc;
}

}  // namespace

extern "C" int cs_vqvae_create(const CsVqvaeConfig* cfg, cs_vqvae** out) {
  if (!cfg || !out) return CS_EINVAL;
  cs_vqvae* u = new (std::nothrow) cs_vqvae();
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

extern "C" void cs_vqvae_destroy(cs_vqvae* u) { delete u; }
extern "C" int cs_vqvae_param_count(const cs_vqvae* u) { return u ? (int)u->params.size() : 0; }
extern "C" int cs_vqvae_param_info(const cs_vqvae* u, int i, const char** name, int64_t shape5[5], int* ndim,
                                   int64_t* raw_offset_bytes) {
  return plan_param_info(u, i, name, shape5, ndim, raw_offset_bytes);
}
extern "C" int64_t cs_vqvae_raw_bytes(const cs_vqvae* u) { return u ? u->raw_bytes : 0; }
extern "C" int64_t cs_vqvae_arena_bytes(const cs_vqvae* u) { return u ? u->arena_bytes : 0; }
extern "C" int cs_vqvae_pack(cs_vqvae* u, const void* raw_dev, void* arena_dev, cs_stream_t stream) {
  const int rc = pack_plan(u, raw_dev, arena_dev, stream);
  if (rc != CS_OK || !u || u->math != CS_MATH_F16X3 || u->extra_slots < 4) return rc;
  // r6: the attention block's static-bound statistics (vqvae.py::_pack: the same kernel on the same tensors, the same values):
  // {max row 2-norm, max |.|} over the q, k, v weights into slots 0-1, over their biases into slots 2-3
  hipStream_t st = (hipStream_t)stream;
  const char* raw = reinterpret_cast<const char*>(raw_dev);
  float* o = reinterpret_cast<float*>(reinterpret_cast<char*>(arena_dev) + u->amax_off) + u->extra_slot0;
  const int c = u->block_in0;
  for (int j = 0; j < 3; ++j) {
    int r2 = cs_weight_rowstats(reinterpret_cast<const float*>(raw + u->params[u->p_qkv_w[j]].raw_off), c, c, o, stream);
    if (r2 == CS_OK)
      r2 = cs_weight_rowstats(reinterpret_cast<const float*>(raw + u->params[u->p_qkv_b[j]].raw_off), 1, c, o + 2, stream);
    if (r2 != CS_OK) return r2;
  }
  float host[4];
  if (hipMemcpyAsync(host, o, sizeof(host), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
    return CS_EINVAL;
  u->attn_l2max = host[0];
  u->attn_bmax = host[3];
  return CS_OK;
}

// Objects decode independently; a 64^3 x 128-channel activation is 134 MB per object (4.3 GB at 32), so a large
// batch is decoded in slices of this many objects, which bounds the workspace.
constexpr int MAX_DECODE_BATCH = 16;

extern "C" int64_t cs_vqvae_workspace_bytes(const cs_vqvae* u, int nb) {
  if (!u || nb <= 0) return CS_EINVAL;
  VExec e(*u, nullptr, nullptr, 0, true, nullptr);
  const int rc = decode(e, nullptr, nullptr, nullptr, nb < MAX_DECODE_BATCH ? nb : MAX_DECODE_BATCH, 1);
  return rc != CS_OK ? rc : e.peak;
}

extern "C" int cs_vqvae_decode(const cs_vqvae* u, const void* arena, const float* latent_ncdhw, float* sdf_ncdhw,
                               int64_t* code_indices, int nb, int quantize, int32_t* status, void* workspace,
                               int64_t workspace_bytes, cs_stream_t stream) {
  if (!u || !u->packed || !arena || !latent_ncdhw || !sdf_ncdhw || !workspace || nb <= 0) return CS_EINVAL;
  if (((uintptr_t)workspace & 15) || ((uintptr_t)arena & 15)) return CS_EINVAL;
  const CsVqvaeConfig& c = u->cfg;
  const int64_t g3 = (int64_t)u->grid * u->grid * u->grid;
  const int64_t r3 = (int64_t)c.resolution * c.resolution * c.resolution;
  for (int b0 = 0; b0 < nb; b0 += MAX_DECODE_BATCH) {
    const int n = nb - b0 < MAX_DECODE_BATCH ? nb - b0 : MAX_DECODE_BATCH;
    VExec e(*u, arena, workspace, workspace_bytes, false, (hipStream_t)stream);   // slices reuse the workspace in
    e.status = status;
    const int rc = decode(e, latent_ncdhw + b0 * c.embed_dim * g3,              // stream order
                          sdf_ncdhw + b0 * c.out_ch * r3, code_indices ? code_indices + b0 * g3 : nullptr, n, quantize);
    if (rc != CS_OK) return rc;
  }
  return CS_OK;
}
