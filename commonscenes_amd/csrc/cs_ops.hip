// HBM-bound elementwise / gather / scatter kernels of the CommonScenes shape path (gfx950):
// GEGLU gate, strided row copy (channel concat), row-vector add, NCDHW<->NDHWC, timestep embedding,
// fused CFG + DDIM update, VQ nearest-code lookup, scene-graph gather / segment-mean, embedding.
#include "cs_common.h"

namespace {

__global__ __launch_bounds__(256) void geglu_kernel(const float* __restrict__ x, float* __restrict__ o,
                                                    int64_t m, int h, int ldx, int ldo) {
  const int h4 = h >> 2;
  const int64_t total = m * h4;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % h4);
    const int64_t row = i / h4;
    const float4 a = *reinterpret_cast<const float4*>(x + row * ldx + c4 * 4);
    const float4 g = *reinterpret_cast<const float4*>(x + row * ldx + h + c4 * 4);
    float4 r;
    r.x = a.x * cs_gelu(g.x);
    r.y = a.y * cs_gelu(g.y);
    r.z = a.z * cs_gelu(g.z);
    r.w = a.w * cs_gelu(g.w);
    *reinterpret_cast<float4*>(o + row * ldo + c4 * 4) = r;
  }
}

__global__ __launch_bounds__(256) void copy_rows_kernel(const float* __restrict__ s, float* __restrict__ d,
                                                        int64_t m, int c, int lds, int ldd) {
  const int c4n = c >> 2;
  const int64_t total = m * c4n;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % c4n);
    const int64_t row = i / c4n;
    *reinterpret_cast<float4*>(d + row * ldd + c4 * 4) =
        *reinterpret_cast<const float4*>(s + row * lds + c4 * 4);
  }
}

__global__ __launch_bounds__(256) void add_rowvec_kernel(float* __restrict__ x, const float* __restrict__ v,
                                                         int64_t m, int c, int ldx, int ldv, int rows) {
  const int c4n = c >> 2;
  const int64_t total = m * c4n;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % c4n);
    const int64_t row = i / c4n;
    float4 a = *reinterpret_cast<float4*>(x + row * ldx + c4 * 4);
    const float4 b = *reinterpret_cast<const float4*>(v + (row / rows) * ldv + c4 * 4);
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    *reinterpret_cast<float4*>(x + row * ldx + c4 * 4) = a;
  }
}

// [nb][c][s] -> [nb][s][cpad]
__global__ __launch_bounds__(256) void nchw_to_ndhwc_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                            int nb, int c, int s, int cpad) {
  const int64_t total = (int64_t)nb * s * cpad;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int ch = (int)(i % cpad);
    const int64_t t = i / cpad;
    const int sp = (int)(t % s);
    const int n = (int)(t / s);
    y[i] = ch < c ? x[((int64_t)n * c + ch) * s + sp] : 0.f;
  }
}

// [nb][s][ldx] -> [nb][c][s]
__global__ __launch_bounds__(256) void ndhwc_to_nchw_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                            int nb, int c, int s, int ldx) {
  const int64_t total = (int64_t)nb * c * s;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int sp = (int)(i % s);
    const int64_t t = i / s;
    const int ch = (int)(t % c);
    const int n = (int)(t / c);
    y[i] = x[((int64_t)n * s + sp) * ldx + ch];
  }
}

__global__ void timestep_embedding_kernel(const int64_t* __restrict__ t, float* __restrict__ out, int nb,
                                          int dim, float max_period) {
  const int half = dim / 2;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nb * dim) return;
  const int b = i / dim, k = i - b * dim;
  float v = 0.f;
  if (k < 2 * half) {
    const int kk = k < half ? k : k - half;
    // freqs = exp(-log(max_period) * arange(half) / half), computed in fp32 like the reference
    const float f = expf(-logf(max_period) * (float)kk / (float)half);
    const float a = (float)t[b] * f;
    v = k < half ? cosf(a) : sinf(a);
  }
  out[i] = v;
}

// x_prev may alias x (in-place update, used by the graph-replay sampler): neither is __restrict__, and every element
// is read before it is written by the same thread.
__global__ __launch_bounds__(256) void ddim_update_kernel(const float* x,
                                                          const float* __restrict__ eps,
                                                          const float* __restrict__ noise,
                                                          float* x_prev,
                                                          float* __restrict__ pred_x0, int64_t n,
                                                          int64_t uc_off, float sqrt_at,
                                                          float sqrt_aprev, float dir_coef,
                                                          float sigma_t, float sqrt_one_minus_at,
                                                          float cfg_scale, int cfg) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    float e;
    if (cfg) {
      const float eu = eps[i];
      const float ec = eps[uc_off + i];
      e = eu + cfg_scale * (ec - eu);
    } else {
      e = eps[i];
    }
    const float xv = x[i];
    const float p0 = (xv - sqrt_one_minus_at * e) / sqrt_at;
    const float dir = dir_coef * e;
    float xp = sqrt_aprev * p0 + dir;
    xp += noise ? sigma_t * noise[i] : 0.f;  // reference adds sigma_t*noise (== +0 when eta == 0)
    if (pred_x0) pred_x0[i] = p0;
    x_prev[i] = xp;
  }
}

// Same update with the five step coefficients read from device memory: a captured HIP graph of one sampling
// step is replayed for every timestep, only the 20-byte coefficient block (and the timestep vector) change.
__global__ __launch_bounds__(256) void ddim_update_dev_kernel(const float* x,
                                                              const float* __restrict__ eps,
                                                              const float* __restrict__ noise,
                                                              float* x_prev,
                                                              float* __restrict__ pred_x0, int64_t n,
                                                              int64_t uc_off, const float* __restrict__ coef,
                                                              float cfg_scale, int cfg) {
  const float sqrt_at = coef[0], sqrt_aprev = coef[1], dir_coef = coef[2], sigma_t = coef[3],
              sqrt_one_minus_at = coef[4];
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    float e;
    if (cfg) {
      const float eu = eps[i];
      const float ec = eps[uc_off + i];
      e = eu + cfg_scale * (ec - eu);
    } else {
      e = eps[i];
    }
    const float xv = x[i];
    const float p0 = (xv - sqrt_one_minus_at * e) / sqrt_at;
    const float dir = dir_coef * e;
    float xp = sqrt_aprev * p0 + dir;
    xp += noise ? sigma_t * noise[i] : 0.f;
    if (pred_x0) pred_x0[i] = p0;
    x_prev[i] = xp;
  }
}

// VQ: codebook (+ squared norms) staged in LDS; every lane scans all codes with broadcast reads.
__global__ __launch_bounds__(256) void vq_kernel(const float* __restrict__ z, const float* __restrict__ cb,
                                                 int64_t* __restrict__ idx, float* __restrict__ zq,
                                                 int64_t m, int ncode, int edim, int ldz, int ldq) {
  extern __shared__ __attribute__((aligned(16))) float sm[];  // [ncode][4] : e0,e1,e2|0,ee
  for (int i = threadIdx.x; i < ncode; i += blockDim.x) {
    float e[3] = {0.f, 0.f, 0.f};
    float ee = 0.f;
    for (int d = 0; d < edim && d < 3; ++d) e[d] = cb[(int64_t)i * edim + d];
    // torch.sum(w**2, dim=1): sequential fp32 sum over edim entries
    for (int d = 0; d < edim; ++d) {
      const float w = cb[(int64_t)i * edim + d];
      ee += w * w;
    }
    sm[4 * i + 0] = e[0];
    sm[4 * i + 1] = e[1];
    sm[4 * i + 2] = e[2];
    sm[4 * i + 3] = ee;
  }
  __syncthreads();
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < m;
       r += (int64_t)gridDim.x * blockDim.x) {
    float zz[3] = {0.f, 0.f, 0.f};
    float z2 = 0.f;
    for (int d = 0; d < edim; ++d) {
      zz[d] = z[r * ldz + d];
      z2 += zz[d] * zz[d];
    }
    float best = INFINITY;
    int bi = 0;
    for (int i = 0; i < ncode; ++i) {
      const float4 c = *reinterpret_cast<const float4*>(sm + 4 * i);
      // z.e as the K=3 dot product of the reference einsum: ((z0*e0) + z1*e1) + z2*e2 via fma chain
      float dot = zz[0] * c.x;
      dot = fmaf(zz[1], c.y, dot);
      dot = fmaf(zz[2], c.z, dot);
      const float d = (z2 + c.w) - 2.0f * dot;
      if (d < best) {
        best = d;
        bi = i;
      }
    }
    idx[r] = bi;
    for (int d = 0; d < edim; ++d) zq[r * ldq + d] = cb[(int64_t)bi * edim + d];
  }
}

__global__ __launch_bounds__(256) void gcn_gather_cat_kernel(const float* __restrict__ obj,
                                                             const float* __restrict__ pred,
                                                             const int64_t* __restrict__ edges,
                                                             float* __restrict__ out, int n_obj, int n_tri,
                                                             int d_obj, int d_pred, int32_t* err) {
  const int width = 2 * d_obj + d_pred;
  const int64_t total = (int64_t)n_tri * width;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % width);
    const int t = (int)(i / width);
    float v;
    if (c < d_obj) {
      const int64_t s = edges[2 * t];
      if (s < 0 || s >= n_obj) { if (err) *err = 1; v = 0.f; }
      else v = obj[s * d_obj + c];
    } else if (c < d_obj + d_pred) {
      v = pred[(int64_t)t * d_pred + (c - d_obj)];
    } else {
      const int64_t o = edges[2 * t + 1];
      if (o < 0 || o >= n_obj) { if (err) *err = 1; v = 0.f; }
      else v = obj[o * d_obj + (c - d_obj - d_pred)];
    }
    out[i] = v;
  }
}

// One thread per (object, channel); walks the edge list in order: all subject contributions, then
// all object contributions -- the summation order of two sequential scatter_add calls on the CPU.
__global__ __launch_bounds__(256) void gcn_segment_mean_kernel(const float* __restrict__ nt,
                                                               const int64_t* __restrict__ edges,
                                                               float* __restrict__ pooled, int n_obj,
                                                               int n_tri, int h, int off_o, int ld_t,
                                                               int32_t* err) {
  const int64_t total = (int64_t)n_obj * h;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % h);
    const int o = (int)(i / h);
    float acc = 0.f;
    float cnt = 0.f;
    for (int t = 0; t < n_tri; ++t) {
      const int64_t s = edges[2 * t];
      if (s < 0 || s >= n_obj) { if (err) *err = 1; continue; }
      if (s == o) {
        acc += nt[(int64_t)t * ld_t + c];
        cnt += 1.f;
      }
    }
    for (int t = 0; t < n_tri; ++t) {
      const int64_t ob = edges[2 * t + 1];
      if (ob < 0 || ob >= n_obj) { if (err) *err = 1; continue; }
      if (ob == o) {
        acc += nt[(int64_t)t * ld_t + off_o + c];
        cnt += 1.f;
      }
    }
    pooled[i] = acc / fmaxf(cnt, 1.f);
  }
}

// CSR by destination: the kernel above walks all T edges twice per output ELEMENT,
// O(O T H) per layer -- 40 us at 34 nodes, quadratic at C4's 258-node graphs.  The incidence lists depend on the edges
// only, so they are built once per graph (one wave per node scans the edge list with ballots: O(O T) integer work,
// edge order preserved) and every layer's pooling is then a segmented sum over each node's own list: subject
// contributions in edge order, then object contributions in edge order -- the order of the reference's two sequential
// scatter_add calls (model/graph.py:176-177), so the sums are bit-identical to the CPU's.
//   csr (int32): start[O] | count_s[O] | count_o[O] | cursor | entries[2T]  (a node's entries: its count_s subject
//   edges then its count_o object edges; the placement of the segments in `entries` is arbitrary, their content is not)
__global__ __launch_bounds__(64) void gcn_csr_build_kernel(const int64_t* __restrict__ edges, int32_t* __restrict__ csr,
                                                           int n_obj, int n_tri, int32_t* err) {
  const int o = blockIdx.x, lane = threadIdx.x;
  int32_t* start = csr;
  int32_t* cnt_s = csr + n_obj;
  int32_t* cnt_o = csr + 2 * n_obj;
  int32_t* cursor = csr + 3 * n_obj;
  int32_t* ent = csr + 3 * n_obj + 1;
  int cs = 0, co = 0;
  for (int t0 = 0; t0 < n_tri; t0 += 64) {                      // pass 1: degrees
    const int t = t0 + lane;
    int64_t s = -1, ob = -1;
    if (t < n_tri) {
      s = edges[2 * t], ob = edges[2 * t + 1];
      if (o == 0 && (s < 0 || s >= n_obj || ob < 0 || ob >= n_obj) && err) *err = 1;
    }
    cs += __popcll(__ballot(s == o));
    co += __popcll(__ballot(ob == o));
  }
  int base = 0;
  if (lane == 0) {
    base = atomicAdd(cursor, cs + co);
    start[o] = base;
    cnt_s[o] = cs;
    cnt_o[o] = co;
  }
  base = __shfl(base, 0, 64);
  int ws = base, wo = base + cs;
  for (int t0 = 0; t0 < n_tri; t0 += 64) {                      // pass 2: fill, edge order kept by the ballot prefix
    const int t = t0 + lane;
    int64_t s = -1, ob = -1;
    if (t < n_tri) s = edges[2 * t], ob = edges[2 * t + 1];
    const unsigned long long ms = __ballot(s == o), mo = __ballot(ob == o);
    const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
    if (s == o) ent[ws + __popcll(ms & below)] = t;
    if (ob == o) ent[wo + __popcll(mo & below)] = t;
    ws += __popcll(ms);
    wo += __popcll(mo);
  }
}

__global__ __launch_bounds__(256) void gcn_segment_mean_csr_kernel(const float* __restrict__ nt,
                                                                   const int32_t* __restrict__ csr,
                                                                   float* __restrict__ pooled, int n_obj, int h,
                                                                   int off_o, int ld_t) {
  const int64_t total = (int64_t)n_obj * h;
  const int32_t* ent = csr + 3 * n_obj + 1;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % h);
    const int o = (int)(i / h);
    const int b = csr[o], ns = csr[n_obj + o], no = csr[2 * n_obj + o];
    float acc = 0.f;
    for (int q = 0; q < ns; ++q) acc += nt[(int64_t)ent[b + q] * ld_t + c];
    for (int q = 0; q < no; ++q) acc += nt[(int64_t)ent[b + ns + q] * ld_t + off_o + c];
    pooled[i] = acc / fmaxf((float)(ns + no), 1.f);
  }
}

__global__ __launch_bounds__(256) void embedding_kernel(const float* __restrict__ table,
                                                        const int64_t* __restrict__ idx,
                                                        float* __restrict__ out, int n, int dim, int n_rows,
                                                        int ldo, int32_t* err) {
  const int64_t total = (int64_t)n * dim;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % dim);
    const int r = (int)(i / dim);
    const int64_t k = idx[r];
    float v = 0.f;
    if (k < 0 || k >= n_rows) { if (err) *err = 1; }
    else v = table[k * dim + c];
    out[(int64_t)r * ldo + c] = v;
  }
}

// one wave per row
__global__ __launch_bounds__(256) void log_softmax_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          int m, int c, int ldx, int ldy) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= m) return;
  const float* xr = x + (int64_t)row * ldx;
  float mx = -INFINITY;
  for (int i = lane; i < c; i += 64) mx = fmaxf(mx, xr[i]);
  mx = wave_max(mx);
  float s = 0.f;
  for (int i = lane; i < c; i += 64) s += expf(xr[i] - mx);
  s = wave_sum(s);
  const float ls = logf(s);
  for (int i = lane; i < c; i += 64) y[(int64_t)row * ldy + i] = (xr[i] - mx) - ls;
}

__global__ __launch_bounds__(256) void synth_fill_kernel(float* __restrict__ out, int64_t n, uint64_t base,
                                                         double scale, double offset) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    uint64_t z = (uint64_t)i * 0x9E3779B97F4A7C15ull + base;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    const int64_t k = (int64_t)(z >> 40);
    const double u = (double)(2 * k - (1ll << 24)) / 16777216.0;
    const double v = u * scale;
    out[i] = (float)(v + offset);
  }
}

inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// ---------------------------------------------------------------------------------------------------------
// Thin-output 3x3x3 convs (cout <= 4: the UNet's eps head, the VQ decoder's conv_out) as "taps as columns":
//   Y[m'][o * 27 + t] = sum_c A[m'][c] * W[o][c][t]        -- ONE pointwise GEMM with 27 * cout columns (K = cin)
//   out[m][o] = bias[o] + sum_t Y[m + off_t][o * 27 + t]    -- this kernel; taps that leave the volume contribute 0
// The implicit GEMM spends a 64-column tile on 1-4 real columns (7 TF/s for the decoder's conv_out); this form does
// 27 * cout useful columns of MFMA work per row and is bound by the Y round trip instead.
// Fixed summation order (t = 0 .. 26, then the bias), one thread per output row: bit-reproducible.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tapsum27_kernel(const float* __restrict__ y, const float* __restrict__ bias,
                                                       float* __restrict__ out, int64_t m_total, int D, int H, int W,
                                                       int cout, int ldy, int ldo) {
  const int64_t m = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (m >= m_total) return;
  const int w = (int)(m % W);
  int64_t t0 = m / W;
  const int h = (int)(t0 % H);
  t0 /= H;
  const int d = (int)(t0 % D);
  int64_t off[27];
  bool in[27];
#pragma unroll
  for (int kd = 0; kd < 3; ++kd)
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int t = (kd * 3 + kh) * 3 + kw;
        in[t] = (unsigned)(d + kd - 1) < (unsigned)D && (unsigned)(h + kh - 1) < (unsigned)H &&
                (unsigned)(w + kw - 1) < (unsigned)W;
        off[t] = (m + ((int64_t)(kd - 1) * H + (kh - 1)) * W + (kw - 1)) * ldy + t;
      }
  for (int o = 0; o < cout; ++o) {
    float v[27];
#pragma unroll
    for (int t = 0; t < 27; ++t) v[t] = in[t] ? y[off[t] + o * 27] : 0.f;
    float acc = 0.f;
#pragma unroll
    for (int t = 0; t < 27; ++t) acc += v[t];
    out[m * ldo + o] = bias ? acc + bias[o] : acc;
  }
}

// W[o][c][t] (torch Conv3d (cout, cin, 3, 3, 3)) -> the F16X3 pack of the pointwise weight [ncolp][cin] whose row
// o * 27 + t is W[o][:][t]; rows >= 27 * cout are zero.  Layout as cs_pack_weight_f16x3 with taps = 1.
__global__ __launch_bounds__(256) void pack_tapcol_f16x3_kernel(const float* __restrict__ w, _Float16* __restrict__ wh,
                                                                _Float16* __restrict__ wl, int cout, int cin, int ncolp,
                                                                int kg, float scale) {
  const int64_t total = (int64_t)kg * ncolp * 8;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int j = (int)(i & 7);
    const int n = (int)((i >> 3) % ncolp);
    const int g = (int)((i >> 3) / ncolp);
    const int c = g * 8 + j;
    float v = 0.f;
    if (c < cin && n < cout * 27) {
      const int o = n / 27, t = n - o * 27;
      v = w[((int64_t)o * cin + c) * 27 + t] * scale;
    }
    const _Float16 hh = (_Float16)v;
    wh[i] = hh;
    wl[i] = (_Float16)(v - (float)hh);
  }
}

}  // namespace

extern "C" int cs_geglu(const float* x, float* out, int m, int h, int ldx, int ldo, cs_stream_t stream) {
  if (!x || !out || m <= 0 || h <= 0 || (h & 3) || (ldx & 3) || (ldo & 3) || ldx < 2 * h || ldo < h)
    return CS_EINVAL;
  if (!al16(x) || !al16(out)) return CS_EINVAL;
  CS_LAUNCH(geglu_kernel, dim3(cs_grid_for((int64_t)m * (h >> 2), 256, 256 * 32)), dim3(256), 0,
                     (hipStream_t)stream, x, out, (int64_t)m, h, ldx, ldo);
  CS_CHECK_LAUNCH();
  return CS_OK;
}

extern "C" int cs_copy_rows(const float* src, float* dst, int64_t m, int c, int lds, int ldd,
                            cs_stream_t stream) {
  if (!src || !dst || m <= 0 || c <= 0 || (c & 3) || (lds & 3) || (ldd & 3) || lds < c || ldd < c)
    return CS_EINVAL;
  if (!al16(src) || !al16(dst)) return CS_EINVAL;
  CS_LAUNCH(copy_rows_kernel, dim3(cs_grid_for(m * (c >> 2), 256, 256 * 32)), dim3(256), 0,
                     (hipStream_t)stream, src, dst, m, c, lds, ldd);
  CS_CHECK_LAUNCH();
  return CS_OK;
}

extern "C" int cs_add_rowvec(float* x, const float* v, int64_t m, int c, int ldx, int ldv, int rows,
                             cs_stream_t stream) {
  if (!x || !v || m <= 0 || c <= 0 || rows <= 0 || (c & 3) || (ldx & 3) || (ldv & 3) || ldx < c ||
      ldv < c)
    return CS_EINVAL;
  if (!al16(x) || !al16(v)) return CS_EINVAL;
  CS_LAUNCH(add_rowvec_kernel, dim3(cs_grid_for(m * (c >> 2), 256, 256 * 32)), dim3(256), 0,
                     (hipStream_t)stream, x, v, m, c, ldx, ldv, rows);
  CS_CHECK_LAUNCH();
  return CS_OK;
}

extern "C" int cs_nchw_to_ndhwc(const float* x, float* y, int nb, int c, int s, int cpad,
                                cs_stream_t stream) {
  if (!x || !y || nb <= 0 || c <= 0 || s <= 0 || cpad < c) return CS_EINVAL;
  CS_LAUNCH(nchw_to_ndhwc_kernel, dim3(cs_grid_for((int64_t)nb * s * cpad, 256, 256 * 32)),
                     dim3(256), 0, (hipStream_t)stream, x, y, nb, c, s, cpad);
  CS_CHECK_LAUNCH();
  return CS_OK;
}

extern "C" int cs_ndhwc_to_nchw(const float* x, float* y, int nb, int c, int s, int ldx,
                                cs_stream_t stream) {
  if (!x || !y || nb <= 0 || c <= 0 || s <= 0 || ldx < c) return CS_EINVAL;
  CS_LAUNCH(ndhwc_to_nchw_kernel, dim3(cs_grid_for((int64_t)nb * s * c, 256, 256 * 32)),
                     dim3(256), 0, (hipStream_t)stream, x, y, nb, c, s, ldx);
  CS_CHECK_LAUNCH();
  return CS_OK;
}

extern "C" int cs_timestep_embedding(const int64_t* t, float* out, int nb, int dim, float max_period,
                                     cs_stream_t stream) {
  if (!t || !out || nb <= 0 || dim <= 1 || !(max_period > 1.f)) return CS_EINVAL;
  const int total = nb * dim;
  CS_LAUNCH(timestep_embedding_kernel, dim3((total + 255) / 256), dim3(256), 0,
                     (hipStream_t)stream, t, out, nb, dim, max_period);
  CS_CHECK_LAUNCH();
  return CS_OK;
}

extern "C" int cs_ddim_cfg_update(const float* x, const float* eps, const float* noise, float* x_prev,
                                  float* pred_x0, int64_t nb, int64_t per, float a_t, float a_prev,
                                  float sigma_t, float sqrt_one_minus_at, float cfg_scale, int cfg,
                                  cs_stream_t stream) {
  if (!x || !eps || !x_prev || nb <= 0 || per <= 0) return CS_EINVAL;
  if (!(a_t > 0.f) || a_prev < 0.f) return CS_EINVAL;
  const int64_t n = nb * per;
  // coefficient arithmetic in fp32, as torch does on fp32 scalars-turned-tensors (ddim.py:228-243)
  const float sqrt_at = sqrtf(a_t);
  const float sqrt_aprev = sqrtf(a_prev);
  const float dir_coef = sqrtf(1.0f - a_prev - sigma_t * sigma_t);
  CS_LAUNCH(ddim_update_kernel, dim3(cs_grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, x,
                     eps, noise, x_prev, pred_x0, n, n, sqrt_at, sqrt_aprev, dir_coef, sigma_t,
                     sqrt_one_minus_at, cfg_scale, cfg);
  CS_CHECK_LAUNCH();
  return CS_OK;
}

extern "C" int cs_ddim_coefficients(float a_t, float a_prev, float sigma_t, float sqrt_one_minus_at,
                                    float* coef5_host) {
  if (!coef5_host || !(a_t > 0.f) || a_prev < 0.f) return CS_EINVAL;
  coef5_host[0] = sqrtf(a_t);
  coef5_host[1] = sqrtf(a_prev);
  coef5_host[2] = sqrtf(1.0f - a_prev - sigma_t * sigma_t);
  coef5_host[3] = sigma_t;
  coef5_host[4] = sqrt_one_minus_at;
  return CS_OK;
}

// max |x| of an fp32 tensor folded into *slot by atomicMax of the bits (non-negative floats order like their bit patterns):
// zero the slot first.  r6: the UNet's conv_in reads the RAW latent x_t, whose magnitude no normalisation bounds -- both hosts
// leave its exact max |.| in a bound slot (CsConvGemm.a_bound) so that the F16X3 operand scale follows it instead of the
// constant 16 + overflow flag + fp32 re-run (openai_model_3d.py:752-766: h = module(h, emb, context) over input_blocks[0]).
__global__ __launch_bounds__(256) void absmax_slot_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ slot) {
  __shared__ float red[4];
  float m = 0.f;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    m = fmaxf(m, fabsf(x[i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    if (!(m < 3.0e38f)) m = 3.0e38f;                 // inf / NaN: a bound no scale can honour -- the kernel's flag reports it
    atomicMax(reinterpret_cast<unsigned int*>(slot), __float_as_uint(m));
  }
}

extern "C" int cs_absmax(const float* x, int64_t n, float* slot, cs_stream_t stream) {
  if (!x || !slot || n <= 0) return CS_EINVAL;
  // (eight elements per thread, at most one workgroup per CU: the r6 first version's 1536 workgroups at 32 objects spent 26 us
  // mostly on their 1536 atomics to one address)
  CS_LAUNCH(absmax_slot_kernel, dim3(cs_grid_for((n + 7) / 8, 256, 256)), dim3(256), 0, (hipStream_t)stream, x, n, slot);
  CS_CHECK_LAUNCH();
  return CS_OK;
}

extern "C" int cs_ddim_cfg_update_dev(const float* x, const float* eps, const float* noise, float* x_prev,
                                      float* pred_x0, int64_t nb, int64_t per, const float* coef5_dev,
                                      float cfg_scale, int cfg, cs_stream_t stream) {
  if (!x || !eps || !x_prev || !coef5_dev || nb <= 0 || per <= 0) return CS_EINVAL;
  const int64_t n = nb * per;
  CS_LAUNCH(ddim_update_dev_kernel, dim3(cs_grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, x, eps, noise,
            x_prev, pred_x0, n, n, coef5_dev, cfg_scale, cfg);
  CS_CHECK_LAUNCH();
  return CS_OK;
}

// PLMS step (samplers/plms.py:175-236): guidance combine, the pseudo linear multistep combination of the current
// and up to three earlier noise predictions, then the same x0 / x_{t-1} update as DDIM at eta = 0 -- one pass.
// The multistep forms are evaluated exactly as the reference writes them (fp32, no contraction).
__global__ __launch_bounds__(256) void plms_update_kernel(const float* x, const float* __restrict__ eps,
                                                          const float* __restrict__ h1, const float* __restrict__ h2,
                                                          const float* __restrict__ h3, float* __restrict__ e_out,
                                                          float* x_prev, float* __restrict__ pred_x0, int64_t n,
                                                          int64_t uc_off, int mode, float sqrt_at, float sqrt_aprev,
                                                          float dir_coef, float sqrt_one_minus_at, float cfg_scale,
                                                          int cfg) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    float e;
    if (cfg) {
      const float eu = eps[i];
      const float ec = eps[uc_off + i];
      e = eu + cfg_scale * (ec - eu);
    } else {
      e = eps[i];
    }
    if (e_out) e_out[i] = e;
    float ep;
    switch (mode) {
      case CS_PLMS_AB2: ep = (3.f * e - h1[i]) / 2.f; break;                                       // plms.py:224
      case CS_PLMS_AB3: ep = (23.f * e - 16.f * h1[i] + 5.f * h2[i]) / 12.f; break;                // :227
      case CS_PLMS_AB4: ep = (55.f * e - 59.f * h1[i] + 37.f * h2[i] - 9.f * h3[i]) / 24.f; break;  // :230
      case CS_PLMS_EULER_AVG: ep = (h1[i] + e) / 2.f; break;                                       // :221, h1 = e_t
      default: ep = e; break;
    }
    const float xv = x[i];
    const float p0 = (xv - sqrt_one_minus_at * ep) / sqrt_at;
    const float xp = sqrt_aprev * p0 + dir_coef * ep;        // + sigma_t * noise with sigma_t == 0 (eta must be 0)
    if (pred_x0) pred_x0[i] = p0;
    x_prev[i] = xp;
  }
}

extern "C" int cs_plms_update(const float* x, const float* eps, const float* h1, const float* h2, const float* h3,
                              float* e_out, float* x_prev, float* pred_x0, int64_t nb, int64_t per, int mode,
                              float a_t, float a_prev, float sqrt_one_minus_at, float cfg_scale, int cfg,
                              cs_stream_t stream) {
  if (!x || !eps || !x_prev || nb <= 0 || per <= 0) return CS_EINVAL;
  if (!(a_t > 0.f) || a_prev < 0.f || mode < CS_PLMS_PLAIN || mode > CS_PLMS_EULER_AVG) return CS_EINVAL;
  const int need = mode == CS_PLMS_AB4 ? 3 : mode == CS_PLMS_AB3 ? 2 : mode == CS_PLMS_PLAIN ? 0 : 1;
  if ((need >= 1 && !h1) || (need >= 2 && !h2) || (need >= 3 && !h3)) return CS_EINVAL;
  const int64_t n = nb * per;
  CS_LAUNCH(plms_update_kernel, dim3(cs_grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, x, eps, h1, h2, h3, e_out,
            x_prev, pred_x0, n, n, mode, sqrtf(a_t), sqrtf(a_prev), sqrtf(1.0f - a_prev), sqrt_one_minus_at, cfg_scale,
            cfg);
  CS_CHECK_LAUNCH();
  return CS_OK;
}

// Nearest-neighbour squared distance of every point of xyz1 in xyz2 (one direction of the Chamfer distance,
// extension/chamfer.cu:11-75 NmDistanceKernel): queries one per lane, the target cloud streamed through LDS in
// 1024-point tiles (every lane reads the same LDS word per step: a broadcast, no bank conflicts).  The distance is
// ((dx*dx + dy*dy) + dz*dz) in fp32 without contraction and the scan keeps the FIRST minimum, as the reference's
// strict `d < best` / `result > best` comparisons do.
__global__ __launch_bounds__(256) void chamfer_nm_kernel(const float* __restrict__ xyz1,
                                                         const float* __restrict__ xyz2, float* __restrict__ dist,
                                                         int32_t* __restrict__ idx, int n, int m) {
  constexpr int TILE = 1024;
  __shared__ float buf[TILE * 3];
  const int b = blockIdx.y;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const float* q = xyz1 + ((int64_t)b * n + (j < n ? j : 0)) * 3;
  const float x1 = q[0], y1 = q[1], z1 = q[2];
  float best = 0.f;
  int best_i = 0;
  const float* t = xyz2 + (int64_t)b * m * 3;
  for (int k0 = 0; k0 < m; k0 += TILE) {
    const int cnt = min(TILE, m - k0);
    __syncthreads();
    for (int e = threadIdx.x; e < cnt * 3; e += blockDim.x) buf[e] = t[(int64_t)k0 * 3 + e];
    __syncthreads();
    for (int k = 0; k < cnt; ++k) {
      const float dx = buf[3 * k] - x1, dy = buf[3 * k + 1] - y1, dz = buf[3 * k + 2] - z1;
      const float d = (dx * dx + dy * dy) + dz * dz;
      if ((k0 + k) == 0 || d < best) {
        best = d;
        best_i = k0 + k;
      }
    }
  }
  if (j < n) {
    dist[(int64_t)b * n + j] = best;
    idx[(int64_t)b * n + j] = best_i;
  }
}

extern "C" int cs_chamfer_nm_distance(const float* xyz1, const float* xyz2, float* dist, int32_t* idx, int b, int n,
                                      int m, cs_stream_t stream) {
  if (!xyz1 || !xyz2 || !dist || !idx || b <= 0 || n <= 0 || m <= 0 || b > 65535) return CS_EINVAL;
  CS_LAUNCH(chamfer_nm_kernel, dim3((n + 255) / 256, b), dim3(256), 0, (hipStream_t)stream, xyz1, xyz2, dist, idx, n,
            m);
  CS_CHECK_LAUNCH();
  return CS_OK;
}

extern "C" int cs_vq_argmin_lookup(const float* z, const float* codebook, int64_t* idx, float* zq,
                                   int64_t m, int ncode, int edim, int ldz, int ldq, cs_stream_t stream) {
  if (!z || !codebook || !idx || !zq || m <= 0 || ncode <= 0 || edim <= 0 || edim > 3 || ldz < edim ||
      ldq < edim)
    return CS_EINVAL;
  const size_t smem = (size_t)ncode * 4 * sizeof(float);
  if (smem > 160 * 1024) return CS_EINVAL;
  if (smem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)vq_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)smem);
    if (e != hipSuccess) return (int)e;
  }
  CS_LAUNCH(vq_kernel, dim3(cs_grid_for(m, 256, 1024)), dim3(256), smem, (hipStream_t)stream, z,
                     codebook, idx, zq, m, ncode, edim, ldz, ldq);
  CS_CHECK_LAUNCH();
  return CS_OK;
}

extern "C" int cs_gcn_gather_cat(const float* obj, const float* pred, const int64_t* edges, float* out,
                                 int n_obj, int n_tri, int d_obj, int d_pred, int32_t* err,
                                 cs_stream_t stream) {
  if (!obj || !pred || !edges || !out || n_obj <= 0 || n_tri <= 0 || d_obj <= 0 || d_pred <= 0)
    return CS_EINVAL;
  CS_LAUNCH(gcn_gather_cat_kernel,
                     dim3(cs_grid_for((int64_t)n_tri * (2 * d_obj + d_pred), 256)), dim3(256), 0,
                     (hipStream_t)stream, obj, pred, edges, out, n_obj, n_tri, d_obj, d_pred, err);
  CS_CHECK_LAUNCH();
  return CS_OK;
}

extern "C" int cs_gcn_segment_mean(const float* new_t, const int64_t* edges, float* pooled, int n_obj,
                                   int n_tri, int h, int off_o, int ld_t, int32_t* err,
                                   cs_stream_t stream) {
  if (!new_t || !edges || !pooled || n_obj <= 0 || n_tri <= 0 || h <= 0 || off_o < 0 || ld_t < off_o + h)
    return CS_EINVAL;
  CS_LAUNCH(gcn_segment_mean_kernel, dim3(cs_grid_for((int64_t)n_obj * h, 256)), dim3(256), 0,
                     (hipStream_t)stream, new_t, edges, pooled, n_obj, n_tri, h, off_o, ld_t, err);
  CS_CHECK_LAUNCH();
  return CS_OK;
}

extern "C" int64_t cs_gcn_csr_ints(int n_obj, int n_tri) {
  return (n_obj <= 0 || n_tri <= 0) ? 0 : 3 * (int64_t)n_obj + 1 + 2 * (int64_t)n_tri;
}

extern "C" int cs_gcn_csr_build(const int64_t* edges, int32_t* csr, int n_obj, int n_tri, int32_t* err,
                                cs_stream_t stream) {
  if (!edges || !csr || n_obj <= 0 || n_tri <= 0) return CS_EINVAL;
  hipError_t e = hipMemsetAsync(csr + 3 * (int64_t)n_obj, 0, sizeof(int32_t), (hipStream_t)stream);   // the cursor
  if (e != hipSuccess) return (int)e;
  CS_LAUNCH(gcn_csr_build_kernel, dim3(n_obj), dim3(64), 0, (hipStream_t)stream, edges, csr, n_obj, n_tri, err);
  CS_CHECK_LAUNCH();
  return CS_OK;
}

extern "C" int cs_gcn_segment_mean_csr(const float* new_t, const int32_t* csr, float* pooled, int n_obj, int h, int off_o,
                                       int ld_t, cs_stream_t stream) {
  if (!new_t || !csr || !pooled || n_obj <= 0 || h <= 0 || off_o < 0 || ld_t < off_o + h) return CS_EINVAL;
  CS_LAUNCH(gcn_segment_mean_csr_kernel, dim3(cs_grid_for((int64_t)n_obj * h, 256)), dim3(256), 0, (hipStream_t)stream,
            new_t, csr, pooled, n_obj, h, off_o, ld_t);
  CS_CHECK_LAUNCH();
  return CS_OK;
}

extern "C" int cs_embedding(const float* table, const int64_t* idx, float* out, int n, int dim,
                            int n_rows, int ldo, int32_t* err, cs_stream_t stream) {
  if (!table || !idx || !out || n <= 0 || dim <= 0 || n_rows <= 0 || ldo < dim) return CS_EINVAL;
  CS_LAUNCH(embedding_kernel, dim3(cs_grid_for((int64_t)n * dim, 256)), dim3(256), 0,
                     (hipStream_t)stream, table, idx, out, n, dim, n_rows, ldo, err);
  CS_CHECK_LAUNCH();
  return CS_OK;
}

// r5: max over the rows of ||row||_2 and max |entry| of a [rows][cols] fp32 matrix (a Linear / 1x1x1 conv weight, a bias or
// a norm's beta as one row) -- what the static operand bounds of a transformer block are built from (cs_transformer_static_
// scales).  One wave per row, fp64 sum of squares, fixed butterfly; out2 = {max row norm, max abs} is combined by atomicMax
// of the (non-negative) bits, so it must be ZERO on entry and several calls may fold into one slot pair.
__global__ __launch_bounds__(256) void weight_rowstats_kernel(const float* __restrict__ w, int rows, int cols,
                                                              float* __restrict__ out2) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* r = w + (int64_t)row * cols;
  double ss = 0.0;
  float am = 0.f;
  for (int c = lane; c < cols; c += 64) {
    const float v = r[c];
    ss += (double)v * v;
    am = fmaxf(am, fabsf(v));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    ss += __shfl_xor(ss, o, 64);
    am = fmaxf(am, __shfl_xor(am, o, 64));
  }
  if (lane == 0) {
    float nrm = (float)sqrt(ss);
    if ((double)nrm * nrm < ss) nrm = __uint_as_float(__float_as_uint(nrm) + 1u);      // round UP: it is a bound
    atomicMax(reinterpret_cast<unsigned int*>(out2), __float_as_uint(nrm));
    atomicMax(reinterpret_cast<unsigned int*>(out2 + 1), __float_as_uint(am));
  }
}

extern "C" int cs_weight_rowstats(const float* w, int rows, int cols, float* out2, cs_stream_t stream) {
  if (!w || !out2 || rows <= 0 || cols <= 0 || ((uintptr_t)out2 & 3)) return CS_EINVAL;
  CS_LAUNCH(weight_rowstats_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, w, rows, cols, out2);
  CS_CHECK_LAUNCH();
  return CS_OK;
}

extern "C" int cs_log_softmax(const float* x, float* y, int m, int c, int ldx, int ldy, cs_stream_t stream) {
  if (!x || !y || m <= 0 || c <= 0 || ldx < c || ldy < c) return CS_EINVAL;
  CS_LAUNCH(log_softmax_kernel, dim3((m + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, y, m, c,
                     ldx, ldy);
  CS_CHECK_LAUNCH();
  return CS_OK;
}

extern "C" int cs_synth_fill(float* out, int64_t n, uint64_t base, double scale, double offset,
                             cs_stream_t stream) {
  if (!out || n <= 0) return CS_EINVAL;
  CS_LAUNCH(synth_fill_kernel, dim3(cs_grid_for(n, 256, 256 * 32)), dim3(256), 0, (hipStream_t)stream,
                     out, n, base, scale, offset);
  CS_CHECK_LAUNCH();
  return CS_OK;
}

extern "C" int cs_tapsum27(const float* y, const float* bias, float* out, int nb, int d, int h, int w, int cout,
                           int ldy, int ldo, cs_stream_t stream) {
  if (!y || !out || nb <= 0 || d <= 0 || h <= 0 || w <= 0 || cout <= 0 || cout > 4) return CS_EINVAL;
  if (ldy < 27 * cout || ldo < cout) return CS_EINVAL;
  const int64_t m = (int64_t)nb * d * h * w;
  if (m * ldy > 0x7fffffffffffLL || (m + 255) / 256 > 0x7fffffffLL) return CS_EINVAL;
  CS_LAUNCH(tapsum27_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, (hipStream_t)stream, y, bias, out, m, d, h,
            w, cout, ldy, ldo);
  CS_CHECK_LAUNCH();
  return CS_OK;
}

extern "C" int cs_pack_weight_f16x3_tapcol(const float* w_torch, void* w_hi, void* w_lo, int cout, int cin, int ncolp,
                                           float scale, cs_stream_t stream) {
  if (!w_torch || !w_hi || !w_lo || cout <= 0 || cout > 4 || cin <= 0 || ncolp < 27 * cout || (ncolp & 3) ||
      !(scale > 0.f))
    return CS_EINVAL;
  const int kg = ((cin + 15) / 16) * 2;
  const int64_t total = (int64_t)kg * ncolp * 8;
  CS_LAUNCH(pack_tapcol_f16x3_kernel, dim3(cs_grid_for(total, 256, 256 * 32)), dim3(256), 0, (hipStream_t)stream,
            w_torch, (_Float16*)w_hi, (_Float16*)w_lo, cout, cin, ncolp, kg, scale);
  CS_CHECK_LAUNCH();
  return CS_OK;
}
