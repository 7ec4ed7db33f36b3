// Shared machinery of the native whole-forward drivers (cs_unet.hip, cs_vqvae.hip): the host-side plan (reference
// parameter table, packed-GEMM recipes, arena layout), the raw -> arena packing pass, and an executor that
// sequences the library's own C entries over a caller-owned workspace (first-fit allocator, dry-run sizing).
// Internal header: everything lives in an anonymous namespace of the including translation unit.
#pragma once
#include "cs_common.h"

#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

namespace {

constexpr int64_t ALIGN = 256;
inline int64_t align_up(int64_t v) { return (v + ALIGN - 1) / ALIGN * ALIGN; }

struct Param {
  std::string name;
  int64_t shape[5];
  int ndim;
  int64_t numel;
  int64_t raw_off;   // bytes into the caller's raw parameter buffer
};

struct Piece {       // rows [row0, row0 + rows) of parameter `param` seen as a [rows_total][cols] matrix;
                     // param < 0: `rows` all-zero rows (channel padding)
  int param;
  int row0;
  int rows;
};

struct Gemm {        // one packed GEMM weight (possibly several reference tensors concatenated along cout)
  std::vector<Piece> w, b;
  int cout = 0, cin = 0, cin_pad = 0, k = 1, taps = 1, ldw = 0;
  // input-channel range [c0, c0 + cin) of source tensors that have src_cin input channels (0 = all of them): the two
  // halves of a channel-split conv (cs_unet.hip: res_block_split); the operand scale is the WHOLE tensor's either way
  int c0 = 0, src_cin = 0;
  // > 0: a thin-output 3x3x3 conv (tap_cout <= 4 output channels) run as "taps as columns": the packed weight is the
  // POINTWISE one with cout = 27 * tap_cout (+ pad) columns (cs_pack_weight_f16x3_tapcol; k = 1 here), and cs_tapsum27
  // adds the 27 shifted columns of every output channel and the bias (ops.py::pack_weight_tapcol)
  int tap_cout = 0;
  int64_t w_off = 0, wlo_off = 0, b_off = -1;
  float acc_scale = 1.f;
  // Upsample's conv folded onto the source grid (cs_conv_gemm_up2): bit 2 / 1 / 0 = D / H / W doubled.  The packed
  // weights are then one image (pair) per output parity class, built from the folded fp32 taps kept at fold_off.
  int up_mask = 0, ncls = 0, fkd = 3, fkh = 3, fkw = 3, amax_slot = -1;
  int64_t fold_off = 0, cls_w_off[8] = {0}, cls_wlo_off[8] = {0};
  float cls_acc_scale[8] = {0};
  // r5: the Winograd-W pack of a 3x3x3 conv beside the direct one (CsConvGemm.a_format = 3, ops.py::pack_weight_wino): four
  // position images, one power-of-two scale; -1 = the geometry can never take that route
  int64_t wino_off = -1, wino_lo_off = -1;
  float wino_acc = 1.f;
  // ... and its F(4,3) sibling (six images): CsConvGemm.a_format = 4
  int64_t wino4_off = -1, wino4_lo_off = -1;
  float wino4_acc = 1.f;
};

struct Norm {
  int gp, bp, c;
  int64_t g_off, b_off;
  float gmax = 1.f, bmax = 0.f;   // max |gamma|, max |beta| (filled by pack_plan in F16X3 mode): norm_a_scale
};

// F16X3 operand scale of a GEMM fed by a normalisation over n elements per statistic: cs_norm_a_scale (csrc/cs_plan.hip),
// the one rule both hosts call
inline float norm_a_scale(float gmax, float bmax, int64_t n) { return cs_norm_a_scale(gmax, bmax, n); }

struct RawCopy {     // a parameter copied verbatim into the arena (e.g. the VQ codebook)
  int param;
  int64_t arena_off;
};

struct FreeBlock {
  int64_t off, size;
};

struct Plan {
  int math = CS_MATH_FP32;
  std::vector<Param> params;
  std::vector<Gemm> gemms;
  std::vector<Norm> norms;
  std::vector<RawCopy> copies;
  int64_t raw_bytes = 0, arena_bytes = 0, amax_off = 0;
  int amax_slots = 0;      // per-parameter |w| maxima, then one per parity class of every folded upsample conv
  int extra_slots = 0;     // ... then this many more floats reserved by the model (cs_unet: 34 per transformer block, r5)
  int extra_slot0 = 0;     // index of the first of them (set by layout_arena)
  bool packed = false;
};

// ---------------------------------------------------------------------------------------------------------
// plan construction
// ---------------------------------------------------------------------------------------------------------
int add_param(Plan& u, const std::string& name, std::initializer_list<int64_t> shape) {
  Param p;
  p.name = name;
  p.ndim = (int)shape.size();
  p.numel = 1;
  int i = 0;
  for (int64_t s : shape) {
    p.shape[i++] = s;
    p.numel *= s;
  }
  for (; i < 5; ++i) p.shape[i] = 1;
  p.raw_off = u.raw_bytes;
  u.raw_bytes += align_up(p.numel * 4);
  u.params.push_back(p);
  return (int)u.params.size() - 1;
}

int add_norm(Plan& u, const std::string& p, int c) {
  Norm n;
  n.gp = add_param(u, p + ".weight", {c});
  n.bp = add_param(u, p + ".bias", {c});
  n.c = c;
  n.g_off = n.b_off = 0;
  u.norms.push_back(n);
  return (int)u.norms.size() - 1;
}

// registers <p>.weight (+ .bias); returns the parameter indices
void add_wb(Plan& u, const std::string& p, int o, int i, int k, bool bias, int& wp, int& bp) {
  if (k > 1)
    wp = add_param(u, p + ".weight", {o, i, k, k, k});
  else if (k == 1)
    wp = add_param(u, p + ".weight", {o, i, 1, 1, 1});
  else
    wp = add_param(u, p + ".weight", {o, i});   // Linear
  bp = bias ? add_param(u, p + ".bias", {o}) : -1;
}

int add_gemm(Plan& u, std::vector<Piece> w, std::vector<Piece> b, int cout, int cin, int k, int cin_pad = 0) {
  Gemm g;
  g.w = std::move(w);
  g.b = std::move(b);
  g.cout = cout;
  g.cin = cin;
  g.k = k < 1 ? 1 : k;
  g.taps = g.k * g.k * g.k;
  g.cin_pad = cin_pad ? cin_pad : (cin + 3) / 4 * 4;
  u.gemms.push_back(g);
  return (int)u.gemms.size() - 1;
}

// conv (k = 3 or 1) or Linear (k = 0) as a single-tensor GEMM
// tapcol: the caller names this layer as a thin-output conv to run as "taps as columns" -- exactly the layers the Python
// hosts name (unet.py: `out.2`, vqvae.py: `decoder.conv_out`); whether it is then taken is ops.py::tapcol_ok's rule.  (r3
// applied the rule to ANY thin 3x3x3 conv here: a config with another one would have made the two hosts sum in a
// different order -- ADVICE r3.)
int add_layer_gemm(Plan& u, const std::string& p, int o, int i, int k, bool bias = true, int cin_pad = 0,
                   int up_mask = 0, bool tapcol = false) {
  int wp, bp;
  add_wb(u, p, o, i, k, bias, wp, bp);
  std::vector<Piece> b;
  if (bp >= 0) b.push_back({bp, 0, o});
  if (tapcol) {
    // cs_tapcol_ok: the one rule (csrc/cs_plan.hip) both hosts ask
    if (!up_mask && cin_pad == 0 && cs_tapcol_ok(o, i, k, u.math)) {
      const int gi = add_gemm(u, {{wp, 0, o}}, b, (27 * o + 3) / 4 * 4, i, 1);
      u.gemms[gi].tap_cout = o;
      return gi;
    }
  }
  const int gi = add_gemm(u, {{wp, 0, o}}, b, o, i, k, cin_pad);
  if (up_mask && k == 3) {      // the conv of an Upsample: folded per output parity class (cs_fold_upsample_weight)
    Gemm& g = u.gemms[gi];
    int32_t n, a, bb, c;
    if (cs_conv_up2_info((up_mask >> 2) & 1, (up_mask >> 1) & 1, up_mask & 1, &n, &a, &bb, &c) == CS_OK) {
      g.up_mask = up_mask;
      g.ncls = n;
      g.fkd = a; g.fkh = bb; g.fkw = c;
    }
  }
  return gi;
}

// input channels [c0, c1) of an already registered conv / linear weight (wp; bias bp or -1) as a GEMM of its own
int add_gemm_cin_range(Plan& u, int wp, int bp, int o, int i_total, int k, int c0, int c1) {
  std::vector<Piece> b;
  if (bp >= 0) b.push_back({bp, 0, o});
  const int gi = add_gemm(u, {{wp, 0, o}}, b, o, c1 - c0, k);
  u.gemms[gi].c0 = c0;
  u.gemms[gi].src_cin = i_total;
  return gi;
}

int add_copy(Plan& u, int param) {
  u.copies.push_back({param, 0});
  return (int)u.copies.size() - 1;
}

void layout_arena(Plan& u) {
  const bool f16 = u.math == CS_MATH_F16X3;
  int64_t off = 0;
  int slots = (int)u.params.size();
  for (Gemm& g : u.gemms) {
    if (g.up_mask) {
      const int ftaps = g.fkd * g.fkh * g.fkw;
      g.ldw = f16 ? g.cout : (g.cout + 3) / 4 * 4;
      g.fold_off = off;
      off += align_up((int64_t)g.ncls * g.cout * g.cin * ftaps * 4);
      for (int c = 0; c < g.ncls; ++c) {
        if (f16) {
          const int64_t img = (int64_t)ftaps * ((g.cin + 15) / 16 * 2) * g.cout * 16;
          g.cls_w_off[c] = off;
          off += align_up(img);
          g.cls_wlo_off[c] = off;
          off += align_up(img);
        } else {
          g.cls_w_off[c] = off;
          off += align_up((int64_t)ftaps * g.cin_pad * g.ldw * 4);
        }
      }
      g.amax_slot = slots;
      slots += g.ncls;
      if (!g.b.empty()) {
        g.b_off = off;
        off += align_up((int64_t)g.cout * 4);
      }
      continue;
    }
    if (f16) {
      const int64_t kg = (int64_t)(g.cin + 15) / 16 * 2;
      const int64_t img = (int64_t)g.taps * kg * g.cout * 16;
      g.ldw = g.cout;
      g.w_off = off;
      off += align_up(img);
      g.wlo_off = off;
      off += align_up(img);
      // (r5: the Winograd-W pack beside the direct one -- also for the input-channel halves of a channel-split conv, scaled
      // by the whole tensor's maximum like their direct packs: ops.py::pack_weight_wino)
      if (g.k == 3 && !g.tap_cout && g.w.size() == 1 && g.w[0].param >= 0 && g.w[0].row0 == 0 && g.w[0].rows == g.cout &&
          (g.cout % 224 == 0 || g.cout % 128 == 0 || g.cout == 64) && g.cin % 8 == 0 && g.cin >= 16 && g.cin_pad == g.cin) {
        const int64_t wimg = 4LL * 9 * kg * g.cout * 16;
        g.wino_off = off;
        off += align_up(wimg);
        g.wino_lo_off = off;
        off += align_up(wimg);
        {
          const int64_t wimg4 = 6LL * 9 * kg * g.cout * 16;
          g.wino4_off = off;
          off += align_up(wimg4);
          g.wino4_lo_off = off;
          off += align_up(wimg4);
        }
      }
    } else {
      g.ldw = (g.cout + 3) / 4 * 4;
      g.w_off = off;
      off += align_up((int64_t)g.taps * g.cin_pad * g.ldw * 4);
    }
    if (!g.b.empty()) {
      g.b_off = off;
      off += align_up((int64_t)g.cout * 4);
    }
  }
  for (Norm& n : u.norms) {
    n.g_off = off;
    off += align_up((int64_t)n.c * 4);
    n.b_off = off;
    off += align_up((int64_t)n.c * 4);
  }
  for (RawCopy& rc : u.copies) {
    rc.arena_off = off;
    off += align_up(u.params[rc.param].numel * 4);
  }
  u.extra_slot0 = slots;
  slots += u.extra_slots;
  u.amax_off = off;
  u.amax_slots = slots;
  off += align_up((int64_t)slots * 4);
  u.arena_bytes = off;
}

int plan_param_info(const Plan* u, int i, const char** name, int64_t shape5[5], int* ndim, int64_t* raw_offset_bytes) {
  if (!u || i < 0 || i >= (int)u->params.size()) return CS_EINVAL;
  const Param& p = u->params[i];
  if (name) *name = p.name.c_str();
  if (shape5) memcpy(shape5, p.shape, sizeof(p.shape));
  if (ndim) *ndim = p.ndim;
  if (raw_offset_bytes) *raw_offset_bytes = p.raw_off;
  return CS_OK;
}

// ---------------------------------------------------------------------------------------------------------
// weight packing kernels (piece-wise versions of cs_pack_weight_f16x3 / cs_relayout_weight: a piece is a row
// range of a reference tensor landing at a column offset of a possibly fused GEMM weight)
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ w, int64_t n, float* __restrict__ out) {
  __shared__ float red[4];
  float m = 0.f;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    m = fmaxf(m, fabsf(w[i]));
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    // non-negative floats order like their bit patterns
    atomicMax(reinterpret_cast<unsigned int*>(out), __float_as_uint(m));
  }
}

__global__ __launch_bounds__(256) void pack_part_f16x3_kernel(const float* __restrict__ w, _Float16* __restrict__ wh,
                                                              _Float16* __restrict__ wl, int rows, int n_off,
                                                              int cout_total, int cin, int taps, int kg_per_tap,
                                                              float scale, int src_cin = 0, int c0 = 0) {
  if (src_cin == 0) src_cin = cin;
  const int64_t total = (int64_t)taps * kg_per_tap * rows * 8;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int j = (int)(i & 7);
    int64_t t = i >> 3;
    const int n = (int)(t % rows);
    t /= rows;
    const int kg = (int)(t % kg_per_tap);
    const int tap = (int)(t / kg_per_tap);
    const int c = kg * 8 + j;
    float v = 0.f;
    if (c < cin) v = w[((int64_t)n * src_cin + c0 + c) * taps + tap] * scale;
    const _Float16 h = (_Float16)v;
    const int64_t o = (((int64_t)tap * kg_per_tap + kg) * cout_total + n_off + n) * 8 + j;
    wh[o] = h;
    wl[o] = (_Float16)(v - (float)h);
  }
}

// r5: the Winograd-W pack of a 3x3x3 conv (or of an input-channel range [c0, c0 + cin) of one): cs_pack_weight_f16x3_wino's
// arithmetic -- u_q over the kw taps formed and split in fp64 -- on a (cout, src_cin, 27) source tensor
__global__ __launch_bounds__(256) void pack_part_f16x3_wino_kernel(const float* __restrict__ w, _Float16* __restrict__ wh,
                                                                   _Float16* __restrict__ wl, int cout, int cin,
                                                                   int kg_per_tap, float scale, int src_cin, int c0,
                                                                   int variant) {
  if (src_cin == 0) src_cin = cin;
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
    const int tap = (int)(t / kg_per_tap);
    const int c = kg * 8 + j;
    double u = 0.0;
    if (c < cin) {
      const float* g = w + ((int64_t)n * src_cin + c0 + c) * 27 + tap * 3;
      const double g0 = g[0], g1 = g[1], g2 = g[2];
      if (variant == 4)
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

__global__ __launch_bounds__(256) void pack_part_f32_kernel(const float* __restrict__ w, float* __restrict__ o,
                                                            int rows, int n_off, int cin, int taps, int cin_pad,
                                                            int ldw, int src_cin = 0, int c0 = 0) {
  if (src_cin == 0) src_cin = cin;
  // w: rows of a (cout, cin, taps) torch tensor; o: [tap][cin_pad][ldw], columns n_off .. n_off + rows
  const int64_t total = (int64_t)taps * cin_pad * rows;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int n = (int)(i % rows);
    const int64_t t = i / rows;
    const int c = (int)(t % cin_pad);
    const int tap = (int)(t / cin_pad);
    float v = 0.f;
    if (c < cin) v = w[((int64_t)n * src_cin + c0 + c) * taps + tap];
    o[((int64_t)tap * cin_pad + c) * ldw + n_off + n] = v;
  }
}

// ---------------------------------------------------------------------------------------------------------
// execution
// ---------------------------------------------------------------------------------------------------------

// raw parameters -> arena: GEMM weights in the layout of the plan's math mode (pieces gathered into fused weights),
// biases, norm affine parameters, verbatim copies.  One stream sync in F16X3 mode (per-tensor |w| maxima).
int pack_plan(Plan* u, const void* raw_dev, void* arena_dev, cs_stream_t stream) {
  if (!u || !raw_dev || !arena_dev) return CS_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const char* raw = (const char*)raw_dev;
  char* arena = (char*)arena_dev;
  const bool f16 = u->math == CS_MATH_F16X3;
  auto src = [&](int param) { return reinterpret_cast<const float*>(raw + u->params[param].raw_off); };
  std::vector<float> amax((size_t)u->amax_slots, 0.f);
  // Upsample convs: fold the 27 taps into the per-parity-class taps first (their maxima are taken over the folded values)
  for (const Gemm& g : u->gemms) {
    if (!g.up_mask) continue;
    if (g.w.size() != 1 || g.w[0].param < 0 || g.w[0].row0 != 0 || g.w[0].rows != g.cout) return CS_EINVAL;
    const int rc = cs_fold_upsample_weight(src(g.w[0].param), reinterpret_cast<float*>(arena + g.fold_off), g.cout, g.cin,
                                           (g.up_mask >> 2) & 1, (g.up_mask >> 1) & 1, g.up_mask & 1, stream);
    if (rc != CS_OK) return rc;
  }
  if (f16) {
    // per-tensor |w| maxima for the power-of-two operand scales: one device pass, ONE host sync (load time)
    float* d_amax = reinterpret_cast<float*>(arena + u->amax_off);
    if (hipMemsetAsync(d_amax, 0, (size_t)u->amax_slots * 4, st) != hipSuccess) return CS_EINVAL;
    for (const Gemm& g : u->gemms) {
        if (g.up_mask) {
        const int64_t n = (int64_t)g.cout * g.cin * g.fkd * g.fkh * g.fkw;
        for (int c = 0; c < g.ncls; ++c) {
          CS_LAUNCH(absmax_kernel, dim3(cs_grid_for(n, 256, 256)), dim3(256), 0, st,
                    reinterpret_cast<const float*>(arena + g.fold_off) + c * n, n, d_amax + g.amax_slot + c);
          CS_CHECK_LAUNCH();
        }
        continue;
      }
      for (const Piece& pc : g.w) {
        if (pc.param < 0) continue;
        const Param& p = u->params[pc.param];
        CS_LAUNCH(absmax_kernel, dim3(cs_grid_for(p.numel, 256, 256)), dim3(256), 0, st, src(pc.param), p.numel,
                  d_amax + pc.param);
        CS_CHECK_LAUNCH();
      }
    }
    for (const Norm& n : u->norms) {
      CS_LAUNCH(absmax_kernel, dim3(1), dim3(256), 0, st, src(n.gp), (int64_t)n.c, d_amax + n.gp);
      CS_CHECK_LAUNCH();
      CS_LAUNCH(absmax_kernel, dim3(1), dim3(256), 0, st, src(n.bp), (int64_t)n.c, d_amax + n.bp);
      CS_CHECK_LAUNCH();
    }
    if (hipMemcpyAsync(amax.data(), d_amax, amax.size() * 4, hipMemcpyDeviceToHost, st) != hipSuccess)
      return CS_EINVAL;
    if (hipStreamSynchronize(st) != hipSuccess) return CS_EINVAL;
    for (Norm& n : u->norms) {
      n.gmax = amax[(size_t)n.gp];
      n.bmax = amax[(size_t)n.bp];
    }
  }
  auto copy_bias = [&](const Gemm& g) -> int {
    int n_off = 0;
    for (const Piece& pc : g.b) {
      if (pc.param < 0) {
        if (hipMemsetAsync(arena + g.b_off + (int64_t)n_off * 4, 0, (size_t)pc.rows * 4, st) != hipSuccess)
          return CS_EINVAL;
      } else if (hipMemcpyAsync(arena + g.b_off + (int64_t)n_off * 4, src(pc.param) + pc.row0, (size_t)pc.rows * 4,
                                hipMemcpyDeviceToDevice, st) != hipSuccess)
        return CS_EINVAL;
      n_off += pc.rows;
    }
    return CS_OK;
  };
  for (Gemm& g : u->gemms) {
    if (g.up_mask) {      // one packed image (pair) per parity class, each with its own power-of-two scale
      const int ftaps = g.fkd * g.fkh * g.fkw;
      const int64_t n = (int64_t)g.cout * g.cin * ftaps;
      for (int c = 0; c < g.ncls; ++c) {
        const float* w = reinterpret_cast<const float*>(arena + g.fold_off) + c * n;
        if (f16) {
          const float m = amax[(size_t)g.amax_slot + c];
          int ex = 0;
          if (m > 0.f && std::isfinite(m)) (void)std::frexp((double)m, &ex);
          const float scale = (float)std::ldexp(1.0, 14 - ex);
          g.cls_acc_scale[c] = 1.0f / (scale * 16.0f);
          const int kg = (g.cin + 15) / 16 * 2;
          const int64_t total = (int64_t)ftaps * kg * g.cout * 8;
          CS_LAUNCH(pack_part_f16x3_kernel, dim3(cs_grid_for(total, 256, 256 * 32)), dim3(256), 0, st, w,
                    (_Float16*)(arena + g.cls_w_off[c]), (_Float16*)(arena + g.cls_wlo_off[c]), g.cout, 0, g.cout,
                    g.cin, ftaps, kg, scale, 0, 0);
        } else {
          if (hipMemsetAsync(arena + g.cls_w_off[c], 0, (size_t)ftaps * g.cin_pad * g.ldw * 4, st) != hipSuccess)
            return CS_EINVAL;
          const int64_t total = (int64_t)ftaps * g.cin_pad * g.cout;
          CS_LAUNCH(pack_part_f32_kernel, dim3(cs_grid_for(total, 256)), dim3(256), 0, st, w,
                    (float*)(arena + g.cls_w_off[c]), g.cout, 0, g.cin, ftaps, g.cin_pad, g.ldw, 0, 0);
        }
        CS_CHECK_LAUNCH();
      }
      const int rc = copy_bias(g);
      if (rc != CS_OK) return rc;
      continue;
    }
    if (g.tap_cout) {     // taps as columns: the whole (tap_cout, cin, 3, 3, 3) tensor -> one pointwise image pair
      if (!f16 || g.w.size() != 1 || g.w[0].param < 0 || g.w[0].row0 != 0 || g.w[0].rows != g.tap_cout) return CS_EINVAL;
      const float m = amax[(size_t)g.w[0].param];
      int ex = 0;
      if (m > 0.f && std::isfinite(m)) (void)std::frexp((double)m, &ex);
      const float scale = (float)std::ldexp(1.0, 14 - ex);
      g.acc_scale = 1.0f / (scale * 16.0f);
      int rc = cs_pack_weight_f16x3_tapcol(src(g.w[0].param), arena + g.w_off, arena + g.wlo_off, g.tap_cout, g.cin,
                                           g.cout, scale, stream);
      if (rc == CS_OK && !g.b.empty()) rc = copy_bias(g);
      if (rc != CS_OK) return rc;
      continue;
    }
    const int cols = (g.src_cin ? g.src_cin : g.cin) * g.taps;    // row length of the reference tensor seen as [cout][cin * taps]
    float scale = 1.f;
    if (f16) {
      // scale by the whole tensor's maximum even when only a row range is used (GEGLU pieces): what
      // ops._pack_weight_f16x3 sees is the permuted full tensor, whose maximum is the same
      float m = 0.f;
      for (const Piece& pc : g.w)
        if (pc.param >= 0) m = fmaxf(m, amax[pc.param]);
      int ex = 0;
      if (m > 0.f && std::isfinite(m)) (void)std::frexp((double)m, &ex);
      scale = (float)std::ldexp(1.0, 14 - ex);
      g.acc_scale = 1.0f / (scale * 16.0f);
      const int64_t img = (int64_t)g.taps * ((g.cin + 15) / 16 * 2) * g.cout * 16;
      bool zero_rows = false;
      for (const Piece& pc : g.w) zero_rows |= pc.param < 0;
      if (zero_rows && (hipMemsetAsync(arena + g.w_off, 0, (size_t)img, st) != hipSuccess ||
                        hipMemsetAsync(arena + g.wlo_off, 0, (size_t)img, st) != hipSuccess))
        return CS_EINVAL;
    } else {
      if (hipMemsetAsync(arena + g.w_off, 0, (size_t)g.taps * g.cin_pad * g.ldw * 4, st) != hipSuccess)
        return CS_EINVAL;
    }
    int n_off = 0;
    for (const Piece& pc : g.w) {
      if (pc.param < 0) {     // zero rows: the images were cleared above
        n_off += pc.rows;
        continue;
      }
      const float* w = src(pc.param) + (int64_t)pc.row0 * cols;
      if (f16) {
        const int kg = (g.cin + 15) / 16 * 2;
        const int64_t total = (int64_t)g.taps * kg * pc.rows * 8;
        CS_LAUNCH(pack_part_f16x3_kernel, dim3(cs_grid_for(total, 256, 256 * 32)), dim3(256), 0, st, w,
                  (_Float16*)(arena + g.w_off), (_Float16*)(arena + g.wlo_off), pc.rows, n_off, g.cout, g.cin,
                  g.taps, kg, scale, g.src_cin, g.c0);
      } else {
        const int64_t total = (int64_t)g.taps * g.cin_pad * pc.rows;
        CS_LAUNCH(pack_part_f32_kernel, dim3(cs_grid_for(total, 256)), dim3(256), 0, st, w,
                  (float*)(arena + g.w_off), pc.rows, n_off, g.cin, g.taps, g.cin_pad, g.ldw, g.src_cin, g.c0);
      }
      CS_CHECK_LAUNCH();
      n_off += pc.rows;
    }
    if (n_off != g.cout) return CS_EINVAL;
    if (f16 && g.wino_off >= 0) {      // r5: the Winograd-W pack (max |u_q| <= 1.5 max |w|: ops.py::pack_weight_wino)
      const int kg = (g.cin + 15) / 16 * 2;
      for (int variant = 2; variant <= (g.wino4_off >= 0 ? 4 : 2); variant += 2) {
        // (max |u_q| <= 1.5 max |w| for F(2,3), <= max |w| for F(4,3): ops.py::pack_weight_wino)
        const double m = (variant == 2 ? 1.5 : 1.0) * (double)amax[(size_t)g.w[0].param];
        int ex = 0;
        if (m > 0.0 && std::isfinite(m)) (void)std::frexp(m, &ex);
        const float wscale = (float)std::ldexp(1.0, 14 - ex);
        (variant == 2 ? g.wino_acc : g.wino4_acc) = 1.0f / (wscale * 16.0f);
        CS_LAUNCH(pack_part_f16x3_wino_kernel, dim3(cs_grid_for((variant + 2LL) * 9 * kg * g.cout * 8, 256, 256 * 32)), dim3(256),
                  0, st, src(g.w[0].param), (_Float16*)(arena + (variant == 2 ? g.wino_off : g.wino4_off)),
                  (_Float16*)(arena + (variant == 2 ? g.wino_lo_off : g.wino4_lo_off)), g.cout, g.cin, kg, wscale, g.src_cin,
                  g.c0, variant);
        CS_CHECK_LAUNCH();
      }
    }
    n_off = 0;
    for (const Piece& pc : g.b) {
      if (pc.param < 0) {
        if (hipMemsetAsync(arena + g.b_off + (int64_t)n_off * 4, 0, (size_t)pc.rows * 4, st) != hipSuccess)
          return CS_EINVAL;
      } else if (hipMemcpyAsync(arena + g.b_off + (int64_t)n_off * 4, src(pc.param) + pc.row0, (size_t)pc.rows * 4,
                                hipMemcpyDeviceToDevice, st) != hipSuccess)
        return CS_EINVAL;
      n_off += pc.rows;
    }
  }
  for (const Norm& n : u->norms) {
    if (hipMemcpyAsync(arena + n.g_off, src(n.gp), (size_t)n.c * 4, hipMemcpyDeviceToDevice, st) != hipSuccess ||
        hipMemcpyAsync(arena + n.b_off, src(n.bp), (size_t)n.c * 4, hipMemcpyDeviceToDevice, st) != hipSuccess)
      return CS_EINVAL;
  }
  for (const RawCopy& rc : u->copies) {
    if (hipMemcpyAsync(arena + rc.arena_off, src(rc.param), (size_t)u->params[rc.param].numel * 4,
                       hipMemcpyDeviceToDevice, st) != hipSuccess)
      return CS_EINVAL;
  }
  u->packed = true;
  return CS_OK;
}

// ---------------------------------------------------------------------------------------------------------
// execution
// ---------------------------------------------------------------------------------------------------------
// per-(row tile, column) GroupNorm partial sums a producing GEMM left in the workspace (CsConvGemm.gn_part; ops.ColStats)
struct Stat {
  int64_t off = -1, bytes = 0;
  int nch = 0, nb = 0, tps = 0, ncls = 1;      // columns, samples the producer ran, tiles per sample (per class), classes
  bool valid() const { return off >= 0; }
};

struct Buf {
  int64_t off = -1, bytes = 0;
  int64_t rows = 0;
  int c = 0;
  // r4: the producers' partials covering this tensor's channels in order -- one segment (a GEMM output) or two (a channel
  // concatenation [h | skip]: seg[1] starts at channel seg[0].nch); released with the buffer
  Stat seg[2];
  int nseg = 0;
  bool half = false;   // fp16 hi image [rows][c] followed by the lo image (the F16X3 GEMMs' pre-split A operand);
                       // same footprint as fp32 [rows][c]
  bool pair = false;   // the INTERLEAVED operand pair (CsConvGemm.a_format = 2): bytes / row stride of fp32 [rows][c],
                       // written by cs_layernorm_pair16
  int wino = 0;        // r5: the Winograd-W operand (CsConvGemm.a_format = 3 / 4), the variant (2 = F(2,3), 4 = F(4,3)): fp16 hi
                       // images [variant + 2][voxels / variant][c] followed by the lo images -- `rows` = (variant + 2) /
                       // variant x the volume's voxels
  float a_scale = 16.f;   // F16X3 operand scale a GEMM reading this buffer uses: the default for activations of unknown
                          // range, the producer's bound for normalisation outputs (norm_a_scale)
};

struct Act {   // an activation volume, channels-last
  Buf b;
  int nb = 0, d = 0, h = 0, w = 0;
};

struct ExecBase {
  const Plan& pl;
  const char* arena;
  char* ws;
  int64_t ws_bytes;
  bool dry;
  hipStream_t st;
  int rc = CS_OK;
  int64_t peak = 0;
  int32_t* status = nullptr;     // caller's sticky CS_STATUS_* word (device) handed to every F16X3 kernel
  bool stats_invariant_only = false;   // (the VQ-VAE decoder) GroupNorm partials only from batch-independent statistics tiles
  // magnitude-bound slots of this forward (CsConvGemm.a_bound; unet.py::_slot / ops.range_bound): one small zeroed region
  Buf amax_arena;
  int amax_next = 0, amax_cap = 0;
  std::vector<FreeBlock> fl;

  ExecBase(const Plan& pl_, const void* arena_, void* ws_, int64_t ws_bytes_, bool dry_, hipStream_t st_)
      : pl(pl_), arena((const char*)arena_), ws((char*)ws_), ws_bytes(ws_bytes_), dry(dry_), st(st_) {
    fl.push_back({0, dry ? (int64_t)1 << 60 : ws_bytes});
  }
  bool ok() const { return rc == CS_OK; }
  void chk(int r) {
    if (rc == CS_OK && r != CS_OK) rc = r;
  }
  float* p(const Buf& b) const { return reinterpret_cast<float*>(ws + b.off); }
  const float* wf(int64_t off) const { return reinterpret_cast<const float*>(arena + off); }

  Buf alloc(int64_t rows, int c) {
    Buf b;
    b.rows = rows;
    b.c = c;
    b.bytes = align_up(rows * c * 4);
    for (size_t i = 0; i < fl.size(); ++i) {
      if (fl[i].size >= b.bytes) {
        b.off = fl[i].off;
        fl[i].off += b.bytes;
        fl[i].size -= b.bytes;
        if (fl[i].size == 0) fl.erase(fl.begin() + i);
        if (b.off + b.bytes > peak) peak = b.off + b.bytes;
        return b;
      }
    }
    chk(CS_ENOMEM);
    b.off = 0;
    return b;
  }
  void release_region(int64_t off, int64_t bytes) {
    Buf t;
    t.off = off;
    t.bytes = bytes;
    release_data(t);
  }
  void amax_begin(int slots) {
    if (pl.math != CS_MATH_F16X3 || cs_debug()->no_dyn_scale || cs_debug()->no_gn_parts) return;
    amax_arena = alloc(slots, 1);
    amax_cap = ok() ? slots : 0;
    amax_next = 0;
    if (ok() && !dry && hipMemsetAsync(ws + amax_arena.off, 0, (size_t)slots * 4, st) != hipSuccess) chk(CS_EINVAL);
  }
  // split-K arrival counters of this forward (CsConvGemm.splitk_sync, ops.py::sync_words): zeroed with the bound slots'
  // region (one memset), returned to zero by the kernels that use them
  Buf sync_arena;
  static constexpr int SYNC_WORDS = 8192;
  void sync_begin() {
    if (pl.math != CS_MATH_F16X3) return;
    sync_arena = alloc(SYNC_WORDS, 1);
    if (ok() && !dry && hipMemsetAsync(ws + sync_arena.off, 0, (size_t)SYNC_WORDS * 4, st) != hipSuccess) chk(CS_EINVAL);
  }
  void set_sync(CsConvGemm& q) const {
    if (q.splitk > 1 && sync_arena.off >= 0 && !dry) {
      q.splitk_sync = reinterpret_cast<int32_t*>(ws + sync_arena.off);
      q.splitk_sync_words = SYNC_WORDS;
    }
  }
  int64_t amax_slot() {       // a fresh slot's workspace offset, or -1 (feature off / arena exhausted)
    if (amax_next >= amax_cap) return -1;
    return amax_arena.off + 4 * (int64_t)amax_next++;
  }
  Stat alloc_stat(int64_t tiles, int nch, int nb, int tps, int ncls) {
    Buf t = alloc(tiles * nch * 4, 1);          // [tiles][nch][2] doubles = 16 bytes per (tile, column)
    Stat st;
    st.off = t.off;
    st.bytes = t.bytes;
    st.nch = nch; st.nb = nb; st.tps = tps; st.ncls = ncls;
    return st;
  }
  // a private copy of a producer's partials (the guidance split: the shared tensor lives on as a skip, its duplicate feeds
  // the next GroupNorm; each copy is released with its buffer)
  Stat dup_stat(const Stat& a) {
    if (!a.valid()) return Stat();
    Stat st = alloc_stat(a.bytes / ((int64_t)a.nch * 16), a.nch, a.nb, a.tps, a.ncls);
    if (ok() && !dry && hipMemcpyAsync(ws + st.off, ws + a.off, (size_t)a.bytes, hipMemcpyDeviceToDevice, this->st) != hipSuccess)
      chk(CS_EINVAL);
    return st;
  }
  void release(Buf& b) {
    for (int i = 0; i < 2; ++i)
      if (b.seg[i].valid()) {
        release_region(b.seg[i].off, b.seg[i].bytes);
        b.seg[i] = Stat();
      }
    b.nseg = 0;
    release_data(b);
  }
  void release_data(Buf& b) {
    if (b.off < 0 || b.bytes == 0) return;
    size_t i = 0;
    while (i < fl.size() && fl[i].off < b.off) ++i;
    fl.insert(fl.begin() + i, {b.off, b.bytes});
    if (i + 1 < fl.size() && fl[i].off + fl[i].size == fl[i + 1].off) {
      fl[i].size += fl[i + 1].size;
      fl.erase(fl.begin() + i + 1);
    }
    if (i > 0 && fl[i - 1].off + fl[i - 1].size == fl[i].off) {
      fl[i - 1].size += fl[i].size;
      fl.erase(fl.begin() + i);
    }
    b.off = -1;
    b.bytes = 0;
  }

  // conv (k^3 taps, stride (1,s,s), nearest upsample (0,up,up)) or pointwise/linear GEMM with the fused epilogue
  // want_stats: the result feeds a GroupNorm -- where the launch can (cs_conv_gemm_epilogue_caps: the one rule both hosts
  // ask), its epilogue leaves the per-(row tile, column) partial sums and the returned buffer carries them (seg[0])
  Buf gemm(const Buf& x, int gi, int nb, int d, int h, int w, int s_hw = 1, int up_hw = 0, int act = CS_ACT_NONE,
           const float* rowvec = nullptr, int ldrv = 0, int rv_rows = 1, const float* res = nullptr, int ldr = 0,
           int tile = 0, int s_d = 1, int up_d = 0, bool want_stats = false, float out_pair = 0.f,
           int64_t a_bound_off = -1) {
    const Gemm& g = pl.gemms[gi];
    const bool tc = g.tap_cout > 0;      // taps as columns: the pointwise GEMM below, then cs_tapsum27
    if (tc) {
      if (s_hw != 1 || s_d != 1 || up_hw || up_d || act != CS_ACT_NONE || rowvec || res || tile || x.half) {
        chk(CS_EINVAL);
        return Buf();
      }
      tile = cs_tapcol_tile((int64_t)nb * d * h * w, g.cout);
    }
    const int k = g.k, pad = k / 2;
    const int vh = h << up_hw, vw = w << up_hw;
    const int dout = ((d << up_d) + 2 * pad - k) / s_d + 1;
    const int hout = (vh + 2 * pad - k) / s_hw + 1;
    const int wout = (vw + 2 * pad - k) / s_hw + 1;
    const int64_t mo = (int64_t)nb * dout * hout * wout;
    const int ocols = act == CS_ACT_GEGLU ? g.cout / 2 : g.cout;
    Buf out = alloc(mo, ocols);
    if (!ok()) return out;
    if (x.c != g.cin_pad || x.rows != (x.wino ? wino_rows(x.wino, (int64_t)nb * d * h * w) : (int64_t)nb * d * h * w) ||
        (x.wino && ((x.wino == 4 ? g.wino4_off : g.wino_off) < 0 || k != 3 || s_hw != 1 || s_d != 1 || up_hw || up_d || tile ||
                    a_bound_off >= 0))) {
      chk(CS_EINVAL);
      return out;
    }
    CsConvGemm q;
    memset(&q, 0, sizeof(q));
    // (the operand format is part of what the tile rule looks at: set in the sizing pass too, so that both passes ask
    // cs_conv_gemm_epilogue_caps about the same launch)
    q.a_format = x.wino == 4 ? 4 : x.wino ? 3 : x.half ? 1 : x.pair ? 2 : 0;
    if (!dry) {
      q.x = p(x);
      if (x.half) q.x_lo = reinterpret_cast<const char*>(p(x)) + x.rows * x.c * 2;
      if (x.wino) q.x_lo = reinterpret_cast<const char*>(p(x)) + x.rows * x.c * 2;      // (the hi images' bytes)
      q.out = p(out);
      q.w = reinterpret_cast<const float*>(arena + (x.wino == 4 ? g.wino4_off : x.wino ? g.wino_off : g.w_off));
      if (pl.math == CS_MATH_F16X3) {
        q.w_lo = arena + (x.wino == 4 ? g.wino4_lo_off : x.wino ? g.wino_lo_off : g.wlo_off);
        // g.acc_scale = 1 / (weight scale * 16); powers of two
        q.acc_scale = (x.wino == 4 ? g.wino4_acc : x.wino ? g.wino_acc : g.acc_scale) * (16.0f / x.a_scale);
        q.a_scale = x.a_scale;
      }
      q.bias = (g.b_off >= 0 && !tc) ? wf(g.b_off) : nullptr;
      q.rowvec = rowvec;
      q.res = res;
      q.status = status;
    }
    q.nb = nb; q.din = d; q.hin = h; q.win = w;
    q.dout = dout; q.hout = hout; q.wout = wout;
    q.cin = g.cin_pad; q.cout = g.cout;
    q.lda = x.c; q.ldw = g.ldw; q.ldo = ocols; q.ldr = res ? ldr : 0; q.ldrv = rowvec ? ldrv : 0;
    q.kd = q.kh = q.kw = k;
    q.sd = s_d; q.sh = q.sw = s_hw;
    q.pd = q.ph = q.pw = pad;
    q.ud = up_d; q.uh = q.uw = up_hw;
    q.act = act; q.rv_rows = rv_rows; q.math = pl.math; q.tile = tile;
    // a_bound_off >= 0: x is a RAW activation whose magnitude bound sits in that workspace slot (range_bound / the
    // GroupNorm over x): the kernel derives the operand scale from it instead of the fixed 16 (ops.py: x_bound=)
    if (a_bound_off >= 0 && pl.math == CS_MATH_F16X3 && !x.half && !x.pair && !dry)
      q.a_bound = reinterpret_cast<const float*>(ws + a_bound_off);
    if (g.up_mask) {
      // Upsample's conv on the source grid: one GEMM per output parity class + interleave (cs_conv_gemm_up2)
      if (g.up_mask != ((up_d << 2) | (up_hw << 1) | up_hw) || s_hw != 1 || s_d != 1 || res || rowvec || tile ||
          x.a_scale != 16.f) {
        chk(CS_EINVAL);
        return out;
      }
      const int64_t ub = cs_conv_gemm_up2_ws_bytes(&q);
      if (ub <= 0) {
        chk(CS_EINVAL);
        return out;
      }
      Buf uws = alloc(ub / 4, 1);
      if (!ok()) return out;
      if (want_stats) stats_for(q, out, nb, (int64_t)d * h * w, (int64_t)nb * d * h * w, g.ncls, g.b_off >= 0, 0, 0);
      if (!dry) {
        const void* wc[8];
        const void* wl[8];
        for (int c = 0; c < g.ncls; ++c) {
          wc[c] = arena + g.cls_w_off[c];
          wl[c] = arena + g.cls_wlo_off[c];
        }
        chk(cs_conv_gemm_up2(&q, wc, wl, g.cls_acc_scale, p(uws), st));
      }
      release(uws);
      return out;
    }
    // small batches: few output tiles -> cut the K loop into slices (same plan function the Python host calls)
    int32_t sk = 1;
    int64_t wsb = 0;
    Buf skws;
    if (x.wino) {
      // r5: the four position results (x K slices) live in the workspace; output transform + epilogue in the reduce kernel
      if (cs_conv_wino_plan(&q, &sk, &wsb) != CS_OK) {
        chk(CS_EINVAL);
        return out;
      }
      skws = alloc(wsb / 4, 1);
      if (!ok()) return out;
      q.splitk = sk;
      q.splitk_ws = dry ? nullptr : p(skws);
    } else if (tile == 0 && cs_conv_gemm_plan(&q, &sk, &wsb) == CS_OK && sk > 1) {
      skws = alloc(wsb / 4, 1);          // the dry run sizes the workspace with it too
      if (!ok()) return out;
      q.splitk = sk;
      q.splitk_ws = dry ? nullptr : p(skws);
      set_sync(q);
    }
    if (want_stats && !tc) stats_for(q, out, nb, (int64_t)dout * hout * wout, mo, 1, g.b_off >= 0, ldr, ldrv);
    if (out_pair > 0.f && !tc && pl.math == CS_MATH_F16X3) {
      // the result's only reader is the next F16X3 GEMM: written as the interleaved operand pair where the launch can
      // (ops.py::conv_gemm out_pair=; the sizing pass needs no answer: same bytes either way)
      // (asked in the sizing pass too -- same bytes either way, but the consumer's tile rule looks at its operand format)
      int32_t pair = 0;
      CsConvGemm probe = q;
      if (dry) dry_operands(probe, g.b_off >= 0, ldr, ldrv);
      if (!cs_debug()->no_pair_epilogue && cs_conv_gemm_epilogue_caps(&probe, nullptr, &pair) == CS_OK && pair) {
        q.out_format = 2;
        q.out_scale = out_pair;
        out.pair = true;
        out.a_scale = out_pair;
      }
    }
    if (!dry) chk(cs_conv_gemm(&q, st));
    release(skws);      // stream-ordered: later kernels that reuse the region run after the reduce
    if (tc) {
      Buf o2 = alloc(mo, g.tap_cout);
      if (ok() && !dry)
        chk(cs_tapsum27(p(out), g.b_off >= 0 ? wf(g.b_off) : nullptr, p(o2), nb, d, h, w, g.tap_cout, g.cout, g.tap_cout, st));
      release(out);
      return o2;
    }
    return out;
  }
  // stride-1 conv / pointwise GEMM on explicit operand views: x (+ x_lo for the pre-split pair) with row stride lda,
  // out with row stride ldo -- channel ranges of wider buffers, sample ranges of a batch (res_block_split)
  // wino (r5): x / x_lo are the hi / lo images of a Winograd-W operand (Buf::wino) of the [nb, d, h, w] volume
  void gemm_view(const float* x, const void* x_lo, int lda, int gi, int nb, int d, int h, int w, float* out, int ldo,
                 const float* rowvec = nullptr, int ldrv = 0, int rv_rows = 1, const float* res = nullptr, int ldr = 0,
                 float a_scale = 16.f, int wino = 0) {
    const Gemm& g = pl.gemms[gi];
    CsConvGemm q;
    memset(&q, 0, sizeof(q));
    if (wino && ((wino == 4 ? g.wino4_off : g.wino_off) < 0 || g.k != 3 || pl.math != CS_MATH_F16X3)) {
      chk(CS_EINVAL);
      return;
    }
    if (wino) q.a_format = wino == 4 ? 4 : 3;
    if (!dry) {
      q.x = x;
      if (x_lo) {
        q.x_lo = x_lo;
        if (!wino) q.a_format = 1;
      }
      q.out = out;
      q.w = reinterpret_cast<const float*>(arena + (wino == 4 ? g.wino4_off : wino ? g.wino_off : g.w_off));
      if (pl.math == CS_MATH_F16X3) {
        q.w_lo = arena + (wino == 4 ? g.wino4_lo_off : wino ? g.wino_lo_off : g.wlo_off);
        q.acc_scale = (wino == 4 ? g.wino4_acc : wino ? g.wino_acc : g.acc_scale) * (16.0f / a_scale);
        q.a_scale = a_scale;
      }
      q.bias = g.b_off >= 0 ? wf(g.b_off) : nullptr;
      q.rowvec = rowvec;
      q.res = res;
      q.status = status;
    } else if (x_lo && !wino) {
      q.a_format = 1;           // the plan looks at it
    }
    const int k = g.k, pad = k / 2;
    q.nb = nb; q.din = d; q.hin = h; q.win = w;
    q.dout = d; q.hout = h; q.wout = w;
    q.cin = g.cin_pad; q.cout = g.cout;
    q.lda = lda; q.ldw = g.ldw; q.ldo = ldo; q.ldr = res ? ldr : 0; q.ldrv = rowvec ? ldrv : 0;
    q.kd = q.kh = q.kw = k;
    q.sd = q.sh = q.sw = 1;
    q.pd = q.ph = q.pw = pad;
    q.act = CS_ACT_NONE; q.rv_rows = rv_rows; q.math = pl.math; q.tile = 0;
    int32_t sk = 1;
    int64_t wsb = 0;
    Buf skws;
    if (wino) {
      if (cs_conv_wino_plan(&q, &sk, &wsb) != CS_OK) {
        chk(CS_EINVAL);
        return;
      }
      skws = alloc(wsb / 4, 1);
      if (!ok()) return;
      q.splitk = sk;
      q.splitk_ws = dry ? nullptr : p(skws);
    } else if (cs_conv_gemm_plan(&q, &sk, &wsb) == CS_OK && sk > 1) {
      skws = alloc(wsb / 4, 1);
      if (!ok()) return;
      q.splitk = sk;
      q.splitk_ws = dry ? nullptr : p(skws);
      set_sync(q);
    }
    if (!dry) chk(cs_conv_gemm(&q, st));
    release(skws);
  }

  // ops.py::_epilogue_extras: ask the library what the launch's epilogue can emit and point the descriptor at a fresh
  // partials region (rps = rows per sample the statistics tiles run over, m_rows = the rows they cover in all)
  // the sizing pass carries no pointers, but cs_conv_gemm_epilogue_caps looks at which epilogue terms exist (and at their
  // alignment): aligned stand-ins for exactly the operands the real pass will set, so both passes get the same answer
  static void dry_operands(CsConvGemm& probe, bool bias, int ldr, int ldrv) {
    const float* some = reinterpret_cast<const float*>((uintptr_t)256);
    probe.out = const_cast<float*>(some);
    probe.bias = bias ? some : nullptr;
    probe.res = ldr > 0 ? some : nullptr;
    probe.ldr = ldr;
    probe.rowvec = ldrv > 0 ? some : nullptr;
    probe.ldrv = ldrv;
  }
  void stats_for(CsConvGemm& q, Buf& out, int nb, int64_t rps, int64_t m_rows, int ncls, bool dry_bias, int dry_ldr,
                 int dry_ldrv) {
    if (cs_debug()->no_gn_parts || pl.math != CS_MATH_F16X3) return;
    CsConvGemm probe = q;
    if (dry) dry_operands(probe, dry_bias, dry_ldr, dry_ldrv);
    int32_t rows = 0;
    if (cs_conv_gemm_epilogue_caps(&probe, &rows, nullptr) != CS_OK || rows <= 0) return;
    if (stats_invariant_only) {
      // r5: only where ONE sample's launch picks the same statistics tiles (ops.py::_epilogue_extras, stats="invariant"):
      // the VQ decoder's bit-exact batch invariance
      // (the SAME kernel variant -- tile code, slab width, statistics rows -- at one sample, at the decode slice limit of
      // sixteen and at this batch: ops.py::_epilogue_extras)
      int32_t t0 = 0, s0 = 0;
      if (nb > 16 || cs_conv_gemm_launch_info(&probe, &t0, &s0) != CS_OK) return;
      for (int nbp : {1, 16}) {
        CsConvGemm one = probe;
        one.nb = nbp;
        int32_t r1 = 0, t1 = 0, s1 = 0;
        if (cs_conv_gemm_epilogue_caps(&one, &r1, nullptr) != CS_OK || cs_conv_gemm_launch_info(&one, &t1, &s1) != CS_OK ||
            r1 != rows || t1 != t0 || s1 != s0)
          return;
      }
    }
    const int64_t tiles = (m_rows + rows - 1) / rows;
    Stat sx = alloc_stat((int64_t)ncls * tiles, q.cout, nb, (int)(rps / rows), ncls);
    if (!ok()) return;
    out.seg[0] = sx;
    out.nseg = 1;
    q.gn_part = dry ? nullptr : reinterpret_cast<double*>(ws + sx.off);
    q.gn_ld = q.cout;
    q.gn_rows = rows;
  }
  // do x's segments cover its channels?  (ops.py::stats_segments)
  bool has_parts(const Buf& x) const {
    if (cs_debug()->no_gn_parts || x.nseg < 1) return false;
    int n = 0;
    for (int i = 0; i < x.nseg; ++i) {
      if (!x.seg[i].valid()) return false;
      n += x.seg[i].nch;
    }
    return n == x.c;
  }
  // (mean, rstd) from the producers' partials: ops.py::groupnorm_stats_from_parts
  void seg_array(const Buf& x, CsGnSeg* sg) const {
    int ch0 = 0;
    for (int i = 0; i < x.nseg; ++i) {
      const Stat& a = x.seg[i];
      sg[i].part = reinterpret_cast<const double*>(ws + a.off);
      sg[i].ld = a.nch; sg[i].col0 = 0; sg[i].ch0 = ch0; sg[i].nch = a.nch;
      sg[i].tiles_per_sample = a.tps; sg[i].ncls = a.ncls; sg[i].nb_src = a.nb; sg[i].reserved = 0;
      ch0 += a.nch;
    }
  }
  float* bound_ptr(int64_t off) const { return off >= 0 ? reinterpret_cast<float*>(ws + off) : nullptr; }
  void finalize_parts(const Buf& x, int nb, float eps, int groups, const Buf& stats, int64_t bound_off = -1) {
    if (!ok() || dry) return;
    CsGnSeg sg[2];
    seg_array(x, sg);
    chk(cs_groupnorm_finalize_parts(sg, x.nseg, nb, (int)(x.rows / nb), x.c, groups, eps, p(stats), bound_ptr(bound_off), st));
  }
  // ops.py::range_bound: the magnitude bound of a RAW tensor that no GroupNorm follows, from its producers' partials, into
  // a fresh slot; -1 when x carries none (the consumer then keeps the fixed scale)
  int64_t range_bound(const Buf& x, int nb, int groups = 32) {
    if (x.c % groups) return -1;
    const int64_t off = amax_slot();
    if (off < 0) return -1;
    if (has_parts(x)) {
      if (ok() && !dry) {
        CsGnSeg sg[2];
        seg_array(x, sg);
        chk(cs_groupnorm_finalize_parts(sg, x.nseg, nb, (int)(x.rows / nb), x.c, groups, 1e-5f, nullptr, bound_ptr(off), st));
      }
      return off;
    }
    // no partials (e.g. a folded Upsample conv at a small batch): one statistics pass over the (small) tensor
    Buf wsb = alloc((cs_groupnorm_ws_bytes(nb, groups) + 3) / 4, 1);
    Buf stats = alloc((int64_t)nb * groups * 2, 1);
    if (ok() && !dry)
      chk(cs_groupnorm_stats_bound(p(x), nb, (int)(x.rows / nb), x.c, x.c, groups, 1e-5f, p(wsb), p(stats), bound_ptr(off), st));
    release(wsb);
    release(stats);
    return off;
  }
  // the statistics pass of a tensor without partials; leaves the magnitude bound too when a slot is given
  void stats_pass(const Buf& x, int nb, float eps, int groups, const Buf& wsb, const Buf& stats, int64_t boff) {
    if (!ok() || dry) return;
    const int rows = (int)(x.rows / nb);
    chk(boff >= 0 ? cs_groupnorm_stats_bound(p(x), nb, rows, x.c, x.c, groups, eps, p(wsb), p(stats), bound_ptr(boff), st)
                  : cs_groupnorm_stats(p(x), nb, rows, x.c, x.c, groups, eps, p(wsb), p(stats), st));
  }

  // bound_off (out): the slot x's magnitude bound went to (-1: none -- x carries no partials or the feature is off)
  Buf gn_stats(const Buf& x, int nb, float eps, int groups = 32, int64_t* bound_off = nullptr) {
    if (bound_off) *bound_off = -1;
    if (has_parts(x)) {
      Buf stats = alloc((int64_t)nb * groups * 2, 1);
      const int64_t off = bound_off ? amax_slot() : -1;
      finalize_parts(x, nb, eps, groups, stats, off);
      if (bound_off) *bound_off = off;
      return stats;
    }
    Buf wsb = alloc((cs_groupnorm_ws_bytes(nb, groups) + 3) / 4, 1);
    Buf stats = alloc((int64_t)nb * groups * 2, 1);
    const int64_t off = bound_off ? amax_slot() : -1;
    if (bound_off) *bound_off = off;
    stats_pass(x, nb, eps, groups, wsb, stats, off);
    release(wsb);
    return stats;
  }
  // GroupNorm apply of channels [ch0, ch0 + c) (x already points at channel ch0, row stride ldx) -> a fresh [rows][c]
  // buffer, fp32 or the pre-split pair depending on the consuming conv
  // variant > 0 (r5): emit the Winograd-W operand of the [nb, vd, vh, vw] volume instead (what wants_wino answered);
  // stats_off: first sample's offset (in samples) into `stats`
  Buf gn_apply_range(const float* x, int ldx, int64_t rows_total, int nb, const Buf& stats, int ni, int groups, int cpg,
                     int ch0, int c, int act, int conv_gi, int64_t m_launch, int variant = 0, int vd = 0, int vh = 0, int vw = 0,
                     int64_t stats_off = 0) {
    const Norm& n = pl.norms[ni];
    Buf y = alloc(variant ? wino_rows(variant, rows_total) : rows_total, c);
    const int rows = (int)(rows_total / nb);
    if (pl.math == CS_MATH_F16X3) y.a_scale = norm_a_scale(n.gmax, n.bmax, (int64_t)rows * cpg);
    if (variant) {
      y.wino = variant;
      y.a_scale *= variant == 2 ? 0.5f : 0.0625f;
      if (ok() && !dry) {
        char* vhi = reinterpret_cast<char*>(p(y));
        chk(cs_groupnorm_apply_wino_range(x, p(stats) + stats_off * groups * 2, wf(n.g_off) + ch0, wf(n.b_off) + ch0, vhi,
                                          vhi + y.rows * c * 2, nb, vd, vh, vw, c, ldx, c, groups, cpg, ch0, act, y.a_scale,
                                          variant, status, st));
      }
      return y;
    }
    if (wants_split16(m_launch, conv_gi)) {      // m_launch: rows per launch of the consuming conv
      y.half = true;
      if (ok() && !dry) {
        char* yh = reinterpret_cast<char*>(p(y));
        chk(cs_groupnorm_apply_split16_range(x, p(stats), wf(n.g_off) + ch0, wf(n.b_off) + ch0, yh,
                                             yh + rows_total * c * 2, nb, rows, c, ldx, c, groups, cpg, ch0, act,
                                             y.a_scale, status, st));
      }
      return y;
    }
    if (ok() && !dry)
      chk(cs_groupnorm_apply_range(x, p(stats), wf(n.g_off) + ch0, wf(n.b_off) + ch0, p(y), nb, rows, c, ldx, c, groups,
                                   cpg, ch0, act, st));
    return y;
  }

  Buf linear(const Buf& x, int gi, int act = CS_ACT_NONE, const float* rowvec = nullptr, int ldrv = 0,
             int rv_rows = 1, const float* res = nullptr, int ldr = 0, int tile = 0, float out_pair = 0.f) {
    return gemm(x, gi, (int)x.rows, 1, 1, 1, 1, 0, act, rowvec, ldrv, rv_rows, res, ldr, tile, 1, 0, false, out_pair);
  }

  // does the GroupNorm feeding conv `gi` over m output rows emit the pre-split operand pair?  (cs_conv_wants_split16: the
  // one rule, csrc/cs_plan.hip; bit-identical to the fp32 route either way)
  bool wants_split16(int64_t m, int gi) const {
    if (gi < 0) return false;
    const Gemm& g = pl.gemms[gi];
    return cs_conv_wants_split16(m, g.cin, g.cout, g.k, !g.up_mask && !g.tap_cout && g.cin_pad == g.cin, pl.math) != 0;
  }

  // r6: the F16X3 operand scales of an attention block fed by a GroupNorm (cs_attnblock_static_scales: the one rule; vqvae.py /
  // unet.py call it with the same statistics) -- false: feature off / not F16X3 / no statistics, the constant 16 then
  bool attnblock_scales(int norm, int64_t n_per_group, int c, float w_l2max, float b_absmax, float qk_scale, float out4[4]) const {
    if (pl.math != CS_MATH_F16X3 || cs_debug()->no_static_scales || !(w_l2max > 0.f)) return false;
    const Norm& nm = pl.norms[norm];
    return cs_attnblock_static_scales(nm.gmax, nm.bmax, n_per_group, c, w_l2max, b_absmax, qk_scale, out4) == CS_OK;
  }

  // self-attention over a fused [rows][3c] q | k | v buffer -> a [rows][c]; F16X3: K / V tile images in a scratch buffer
  // where the library has that path (cs_attn_f16x3_ws_bytes > 0)
  // qkv_scales (r5): the static-bound operand scales of q * scale, k, v (cs_transformer_static_scales), or nullptr = 16
  void self_attention(const Buf& qkv, const Buf& a, int nb, int n, int heads, int dh, int c, float scale,
                      const float* qkv_scales = nullptr) {
    Buf ws;
    if (qkv_scales && pl.math == CS_MATH_F16X3) {
      // (r6: the image path too -- cs_attn_selfattn_f16x3_ws_scaled; ops.py::attention does the same)
      const int64_t wsb2 = cs_attn_f16x3_ws_bytes(nb, n, n, heads, dh);
      if (wsb2 > 0) ws = alloc(wsb2 / 4, 1);
      if (ok() && !dry) {
        const float* q = p(qkv);
        chk(wsb2 > 0 ? cs_attn_selfattn_f16x3_ws_scaled(q, q + c, q + 2 * c, p(a), nb, n, n, heads, dh, 3 * c, 3 * c, 3 * c, c,
                                                        scale, qkv_scales[0], qkv_scales[1], qkv_scales[2], status, (void*)p(ws), st)
                     : cs_attn_selfattn_f16x3_scaled(q, q + c, q + 2 * c, p(a), nb, n, n, heads, dh, 3 * c, 3 * c, 3 * c, c, scale,
                                                     qkv_scales[0], qkv_scales[1], qkv_scales[2], status, st));
      }
      if (wsb2 > 0) release(ws);
      return;
    }
    const int64_t wsb = pl.math == CS_MATH_F16X3 ? cs_attn_f16x3_ws_bytes(nb, n, n, heads, dh) : 0;
    if (wsb > 0) ws = alloc(wsb / 4, 1);
    if (ok() && !dry) {
      const float* q = p(qkv);
      chk(pl.math == CS_MATH_F16X3
              ? cs_attn_selfattn_f16x3_ws(q, q + c, q + 2 * c, p(a), nb, n, n, heads, dh, 3 * c, 3 * c, 3 * c, c, scale,
                                          status, wsb > 0 ? (void*)p(ws) : nullptr, st)
              : cs_attn_selfattn(q, q + c, q + 2 * c, p(a), nb, n, n, heads, dh, 3 * c, 3 * c, 3 * c, c, scale, st));
    }
    if (wsb > 0) release(ws);
  }

  // r5: does the GroupNorm feeding the 3x3x3 conv `gi` over an [nb, d, h, w] volume emit the Winograd-W operand?
  // (cs_conv_wino_ok: the one rule; ops.py::wants_wino)
  // (the variant: 0 = direct form, 2 = F(2,3), 4 = F(4,3))
  int wants_wino(int gi, int nb, int d, int h, int w) const {
    if (gi < 0 || d <= 0 || pl.math != CS_MATH_F16X3 || cs_debug()->no_split16) return 0;
    const Gemm& g = pl.gemms[gi];
    if (g.wino_off < 0) return 0;
    CsConvGemm q;
    memset(&q, 0, sizeof(q));
    q.nb = nb; q.din = q.dout = d; q.hin = q.hout = h; q.win = q.wout = w;
    q.cin = g.cin_pad; q.cout = g.cout; q.lda = g.cin_pad; q.ldo = g.cout; q.ldw = g.ldw;
    q.kd = q.kh = q.kw = 3;
    q.sd = q.sh = q.sw = q.pd = q.ph = q.pw = 1;
    q.math = pl.math; q.rv_rows = 1;
    const int v = cs_conv_wino_ok(&q);
    return v == 4 ? (g.wino4_off >= 0 ? 4 : 2) : v;
  }
  // rows of the Buf that holds variant v's operand of a volume of `voxels` rows
  static int64_t wino_rows(int v, int64_t voxels) { return voxels / v * (v + 2); }
  // GroupNorm + activation emitted as the Winograd-W operand (y: 2 * x.rows "rows", see Buf::wino) at HALF the
  // normalisation's operand scale (ops.py::groupnorm wino=True)
  void emit_wino(const Buf& x, const Norm& n, const Buf& stats, Buf& y, int nb, int d, int h, int w, int groups, int act,
                 int variant) {
    y.wino = variant;
    y.a_scale *= variant == 2 ? 0.5f : 0.0625f;      // |transformed| <= 2 x resp. 10 x the activation's bound
    if (ok() && !dry) {
      char* vh = reinterpret_cast<char*>(p(y));
      chk(cs_groupnorm_apply_wino_range(p(x), p(stats), wf(n.g_off), wf(n.b_off), vh, vh + y.rows * x.c * 2, nb, d, h, w, x.c,
                                        x.c, x.c, groups, x.c / groups, 0, act, y.a_scale, variant, status, st));
    }
  }

  // conv_gi: the 3x3x3 conv that consumes the result (decides the output format), or -1
  // bound_off (out, optional): wanted when a consumer will read x RAW (the ResBlock's skip conv): the slot x's magnitude
  // bound went to, -1 if none (ops.py::groupnorm bound=)
  // vd, vh, vw (r5): the volume's extents where the consumer may take the Winograd-W route (the UNet's ResBlocks)
  Buf groupnorm(const Buf& x, int ni, int nb, float eps, int act, int groups = 32, int conv_gi = -1,
                int64_t* bound_off = nullptr, int vd = 0, int vh = 0, int vw = 0) {
    const Norm& n = pl.norms[ni];
    const int wn = wants_wino(conv_gi, nb, vd, vh, vw);
    Buf y = alloc(wn ? wino_rows(wn, x.rows) : x.rows, x.c);
    if (bound_off) *bound_off = -1;
    if (has_parts(x)) {      // r4: statistics from the producers' partials, the tensor is read once (ops.py::groupnorm)
      const int64_t boff = bound_off ? amax_slot() : -1;
      if (bound_off) *bound_off = boff;
      Buf stats = alloc((int64_t)nb * groups * 2, 1);
      if (pl.math == CS_MATH_F16X3) y.a_scale = norm_a_scale(n.gmax, n.bmax, (x.rows / nb) * (int64_t)(x.c / groups));
      const int rows = (int)(x.rows / nb);
      if (wn) {
        finalize_parts(x, nb, eps, groups, stats, boff);
        emit_wino(x, n, stats, y, nb, vd, vh, vw, groups, act, wn);
      } else if (wants_split16(x.rows, conv_gi)) {
        y.half = true;
        finalize_parts(x, nb, eps, groups, stats, boff);
        if (ok() && !dry) {
          char* yh = reinterpret_cast<char*>(p(y));
          chk(cs_groupnorm_apply_split16(p(x), p(stats), wf(n.g_off), wf(n.b_off), yh, yh + x.rows * x.c * 2, nb, rows, x.c,
                                         x.c, x.c, groups, act, y.a_scale, status, st));
        }
      } else if (ok() && !dry) {
        // one call: a single launch for small tensors, finalize + apply otherwise (cs_groupnorm_parts decides)
        CsGnSeg sg[2];
        seg_array(x, sg);
        chk(cs_groupnorm_parts(p(x), sg, x.nseg, wf(n.g_off), wf(n.b_off), p(y), nb, rows, x.c, x.c, x.c, groups, eps, act,
                               p(stats), bound_ptr(boff), st));
      }
      release(stats);
      return y;
    }
    Buf wsb = alloc((cs_groupnorm_ws_bytes(nb, groups) + 3) / 4, 1);
    Buf stats = alloc((int64_t)nb * groups * 2, 1);
    const int64_t boff = bound_off ? amax_slot() : -1;
    if (bound_off) *bound_off = boff;
    if (pl.math == CS_MATH_F16X3) y.a_scale = norm_a_scale(n.gmax, n.bmax, (x.rows / nb) * (int64_t)(x.c / groups));
    if (wn) {
      stats_pass(x, nb, eps, groups, wsb, stats, boff);
      emit_wino(x, n, stats, y, nb, vd, vh, vw, groups, act, wn);
      release(wsb);
      release(stats);
      return y;
    }
    if (wants_split16(x.rows, conv_gi)) {
      y.half = true;
      stats_pass(x, nb, eps, groups, wsb, stats, boff);
      if (ok() && !dry) {
        const int rows = (int)(x.rows / nb);
        char* yh = reinterpret_cast<char*>(p(y));
        chk(cs_groupnorm_apply_split16(p(x), p(stats), wf(n.g_off), wf(n.b_off), yh, yh + x.rows * x.c * 2, nb, rows, x.c,
                                       x.c, x.c, groups, act, y.a_scale, status, st));
      }
      release(wsb);
      release(stats);
      return y;
    }
    if (boff >= 0) {       // bound wanted: statistics (+ bound) and apply as two launches (ops.py::groupnorm)
      stats_pass(x, nb, eps, groups, wsb, stats, boff);
      if (ok() && !dry)
        chk(cs_groupnorm_apply(p(x), p(stats), wf(n.g_off), wf(n.b_off), p(y), nb, (int)(x.rows / nb), x.c, x.c, x.c, groups,
                               act, st));
    } else if (ok() && !dry) {
      const int rows = (int)(x.rows / nb);
      chk(cs_groupnorm(p(x), wf(n.g_off), wf(n.b_off), p(y), nb, rows, x.c, x.c, x.c, groups, eps, act, p(wsb), p(stats), st));
    }
    release(wsb);
    release(stats);
    return y;
  }
  Buf layernorm(const Buf& x, int ni) {
    const Norm& n = pl.norms[ni];
    Buf y = alloc(x.rows, x.c);
    if (pl.math == CS_MATH_F16X3) y.a_scale = norm_a_scale(n.gmax, n.bmax, x.c);
    // LayerNorm outputs only ever feed GEMMs: in F16X3 mode they are written as the interleaved operand pair
    // (ops.py::layernorm pair_scale; CS_NO_PAIR16=1 keeps fp32 for A/B runs)
    if (pl.math == CS_MATH_F16X3 && !cs_debug()->no_pair16 && x.c % 16 == 0) {
      y.pair = true;
      if (ok() && !dry)
        chk(cs_layernorm_pair16(p(x), wf(n.g_off), wf(n.b_off), p(y), (int)x.rows, x.c, x.c, x.c, 1e-5f, y.a_scale,
                                status, st));
      return y;
    }
    if (ok() && !dry) chk(cs_layernorm(p(x), wf(n.g_off), wf(n.b_off), p(y), (int)x.rows, x.c, x.c, x.c, 1e-5f, st));
    return y;
  }
};

}  // namespace
