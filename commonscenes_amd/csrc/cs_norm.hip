// GroupNorm (NDHWC) and LayerNorm for gfx950.  Both are HBM-bound streaming kernels:
// float4 coalesced loads, fp64 accumulation for GroupNorm statistics (a group spans up to
// 4096*42 elements), wave64 shuffles for LayerNorm rows.
#include <cstdlib>

#include "cs_common.h"

namespace {

constexpr int GN_MIN_SPLIT_ROWS = 16;  // minimum rows handled by one statistics block
constexpr int GN_MAX_SPLITS = 256;
constexpr int64_t GN_SMALL_BYTES = 16 << 20;   // single-launch path: whole tensor at most this (L2 / MALL resident)
constexpr int64_t GN_SMALL_GROUP = 11264;      // ... and at most this many elements per (sample, group) workgroup
                                               // (16x4x4 voxels x 42 channels: 14.5 vs 18.0 / 25.4 vs 30.2 us, tools/gn_bench.py)

// Pass 1: partial[n][split][g] = (sum, sumsq) in fp64.  Each thread owns fixed channel chunks so
// its accumulation order is fixed; the cross-thread reduction runs in a fixed order too, so the
// statistics are bit-reproducible run to run.
__global__ __launch_bounds__(256) void gn_partial_kernel(const float* __restrict__ x, int rows, int c,
                                                         int ldx, int groups, int nsplit, int rps,
                                                         double* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) double sm[];  // [rowlanes][c][2]
  const int n = blockIdx.x / nsplit;
  const int split = blockIdx.x % nsplit;
  const int ch4 = c >> 2;
  const int tpr = ch4 < 256 ? ch4 : 256;  // threads per row
  const int rowlanes = 256 / tpr;
  const int tid = threadIdx.x;
  const int rl = tid / tpr;
  const int cl = tid - rl * tpr;
  const int r0 = split * rps;
  const int r1 = min(rows, r0 + rps);
  const float* xb = x + (int64_t)n * rows * ldx;
  if (rl < rowlanes) {
    for (int c4 = cl; c4 < ch4; c4 += tpr) {
      double s0 = 0, s1 = 0, s2 = 0, s3 = 0, q0 = 0, q1 = 0, q2 = 0, q3 = 0;
      // four rows' loads in flight per thread (one 16-byte load per iteration left the kernel at ~45 % of the HBM
      // rate: too few bytes in flight per CU); the accumulation order is unchanged
      for (int r = r0 + rl; r < r1; r += 4 * rowlanes) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int rr = r + u * rowlanes;
          v[u] = rr < r1 ? *reinterpret_cast<const float4*>(xb + (int64_t)rr * ldx + c4 * 4)
                         : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (r + u * rowlanes < r1) {
            s0 += v[u].x; q0 += (double)v[u].x * v[u].x;
            s1 += v[u].y; q1 += (double)v[u].y * v[u].y;
            s2 += v[u].z; q2 += (double)v[u].z * v[u].z;
            s3 += v[u].w; q3 += (double)v[u].w * v[u].w;
          }
        }
      }
      double* d = sm + ((int64_t)rl * c + c4 * 4) * 2;
      d[0] = s0; d[1] = q0; d[2] = s1; d[3] = q1; d[4] = s2; d[5] = q2; d[6] = s3; d[7] = q3;
    }
  }
  __syncthreads();
  if (tid < groups) {
    const int cpg = c / groups;
    double s = 0, q = 0;
    for (int l = 0; l < rowlanes; ++l)
      for (int k = 0; k < cpg; ++k) {
        const double* d = sm + ((int64_t)l * c + tid * cpg + k) * 2;
        s += d[0];
        q += d[1];
      }
    double* o = partial + (((int64_t)n * nsplit + split) * groups + tid) * 2;
    o[0] = s;
    o[1] = q;
  }
}

// One wave per (sample, group): lanes take the row-slice partials in a strided fashion and a fixed butterfly
// combines them, so the result does not depend on scheduling (the serial version walked up to 256 dependent
// loads per thread: 17-27 us for what is a few KB of data).
__device__ __forceinline__ void gn_bound_max(float* bound, double mean, double var, double count);

__global__ __launch_bounds__(256) void gn_finalize_kernel(const double* __restrict__ partial, int nsplit, int groups,
                                                          double count, float eps, float* __restrict__ stats,
                                                          int total, float* __restrict__ bound = nullptr) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);  // n*groups + g
  if (i >= total) return;
  const int n = i / groups, g = i - n * groups;
  double s = 0, q = 0;
  for (int k = lane; k < nsplit; k += 64) {
    const double* d = partial + (((int64_t)n * nsplit + k) * groups + g) * 2;
    s += d[0];
    q += d[1];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o, 64);
    q += __shfl_xor(q, o, 64);
  }
  if (lane == 0) {
    const double mean = s / count;
    double var = q / count - mean * mean;
    if (var < 0) var = 0;
    stats[2 * i] = (float)mean;
    stats[2 * i + 1] = (float)(1.0 / sqrt(var + (double)eps));
    if (bound) gn_bound_max(bound, mean, var, count);
  }
}

// Statistics from the producers' per-(row tile, column) partials (CsConvGemm.gn_part) instead of a pass over the tensor.
// One wave per (sample, group): for each channel of the group in turn the lanes stride over that channel's tiles, then a
// fixed butterfly -- the result does not depend on scheduling or on how many samples share the launch.
struct GnSegs {
  CsGnSeg s[4];
  int n;
};
// the (sum, sum of squares) of group g of sample n from the segments' partials: the lanes of one wave stride over the
// (channel, tile) pairs of each segment's share of the group -- independent loads, all in flight together (a first version
// walked the channels one after the other: up to 42 dependent load latencies per group) -- in a fixed order
__device__ __forceinline__ void gn_group_parts(const GnSegs& sg, int n, int g, int cpg, int lane, int nlanes, double& s,
                                               double& q) {
  s = 0;
  q = 0;
  const int c_lo = g * cpg, c_hi = c_lo + cpg;
#pragma unroll
  for (int si = 0; si < 4; ++si) {
    if (si >= sg.n) break;
    const CsGnSeg& sp = sg.s[si];
    const int lo = max(c_lo, sp.ch0), hi = min(c_hi, sp.ch0 + sp.nch);
    if (hi <= lo) continue;
    const int tps = sp.tiles_per_sample, nt = sp.ncls * tps, items = (hi - lo) * nt;
    const int64_t cls_stride = (int64_t)sp.nb_src * tps;
    const int64_t t0 = (int64_t)(n % sp.nb_src) * tps;
    // four items' loads in flight per lane (the plain loop waited out one load latency per item: 28 items a lane at the
    // 16-row partials of the split-K reduce = 9.6 us for a kernel that moves 30 KB); added in the same order as before
    for (int j = lane; j < items; j += 4 * nlanes) {
      double2 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int ju = j + u * nlanes;
        v[u] = make_double2(0.0, 0.0);
        if (ju < items) {
          const int k = ju / nt, r = ju - k * nt;           // channel lo + k, tile r = cls * tps + t
          const int cls = r / tps, t = r - cls * tps;
          v[u] = *reinterpret_cast<const double2*>(sp.part +
                                                   ((cls * cls_stride + t0 + t) * sp.ld + sp.col0 + (lo + k - sp.ch0)) * 2);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (j + u * nlanes < items) {
          s += v[u].x;
          q += v[u].y;
        }
    }
  }
}

// |mean| + std * sqrt(n - 1) >= max |x| over the group (Samuelson's inequality), rounded UP to fp32; the maximum over all
// (sample, group)s bounds the tensor: what CsConvGemm.a_bound reads.  atomicMax of the bits: order-independent.
__device__ __forceinline__ float gn_bound_value(double mean, double var, double count) {
  const double b = fabs(mean) + sqrt(var * (count > 1.0 ? count - 1.0 : 1.0));
  float bf = (float)b;
  if ((double)bf < b) bf = __uint_as_float(__float_as_uint(bf) + 1u);
  return (bf == bf && bf > 0.f) ? bf : 0.f;
}
__device__ __forceinline__ void gn_bound_commit(float* bound, float bf) {
  if (bf > 0.f) atomicMax(reinterpret_cast<unsigned int*>(bound), __float_as_uint(bf));
}
__device__ __forceinline__ void gn_bound_max(float* bound, double mean, double var, double count) {
  gn_bound_commit(bound, gn_bound_value(mean, var, count));
}

// One word per tensor: the 2048 (sample, group)s of a 32-object launch each hitting it with an atomic serialise in L2
// (measured: 30 us for a kernel that otherwise takes 5), so a workgroup is SIXTEEN waves = sixteen (sample, group)s and
// commits one maximum.
constexpr int GN_FIN_WAVES = 16;
// WPG = waves per (sample, group).  r5: at one or two objects a group of the 16^3 level is 1792 (channel, tile) pairs of
// 16-row partials -- 28 dependent-latency rounds for ONE wave (15.9 us per launch, 27 launches per one-object step); with
// four waves per group the same kernel takes a third of that.  WPG is chosen from the pairs per group alone (a property of
// the sample's own partials, cs_groupnorm_finalize_parts), never from the batch: the summation tree of a sample does not
// depend on what else shares the launch.
template <int WPG>
__global__ __launch_bounds__(64 * GN_FIN_WAVES) void gn_finalize_parts_kernel(const GnSegs sg, int groups, int cpg,
                                                                              double count, float eps,
                                                                              float* __restrict__ stats,
                                                                              float* __restrict__ bound, int total) {
  constexpr int GPB = GN_FIN_WAVES / WPG;           // (sample, group)s per workgroup
  __shared__ float bmax[GN_FIN_WAVES];
  __shared__ double wsum[GN_FIN_WAVES][2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int slot = wave / WPG, sub = wave - slot * WPG;
  const int i = blockIdx.x * GPB + slot;  // n * groups + g
  float bf = 0.f;
  double s = 0, q = 0;
  if (i < total) {
    const int n = i / groups, g = i - n * groups;
    gn_group_parts(sg, n, g, cpg, sub * 64 + lane, 64 * WPG, s, q);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      s += __shfl_xor(s, o, 64);
      q += __shfl_xor(q, o, 64);
    }
  }
  if constexpr (WPG > 1) {
    if (lane == 0) {
      wsum[wave][0] = s;
      wsum[wave][1] = q;
    }
    __syncthreads();
    s = 0;
    q = 0;
#pragma unroll
    for (int w = 0; w < WPG; ++w) {               // fixed order over the group's waves
      s += wsum[slot * WPG + w][0];
      q += wsum[slot * WPG + w][1];
    }
  }
  if (i < total && sub == 0) {
    const double mean = s / count;
    double var = q / count - mean * mean;
    if (var < 0) var = 0;
    if (lane == 0 && stats) {
      stats[2 * i] = (float)mean;
      stats[2 * i + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
    if (bound) bf = gn_bound_value(mean, var, count);
  }
  if (!bound) return;                  // (kernel-uniform)
  if (lane == 0) bmax[wave] = bf;
  __syncthreads();
  if (threadIdx.x == 0) {
    float m = bmax[0];
#pragma unroll
    for (int w = 1; w < GN_FIN_WAVES; ++w) m = fmaxf(m, bmax[w]);
    gn_bound_commit(bound, m);
  }
}

// One workgroup = a run of rows of ONE sample; a thread keeps its float4 column for the whole run, so the group
// statistics and the affine parameters of its four channels are loaded once and the row loop has no index
// arithmetic beyond a pointer bump (the previous flat element loop spent its time in 64-bit divisions: 45 % of the
// HBM rate).  Four rows' loads are in flight per thread.
__global__ __launch_bounds__(256) void gn_apply_kernel(const float* __restrict__ x,
                                                       const float* __restrict__ stats,
                                                       const float* __restrict__ gamma,
                                                       const float* __restrict__ beta,
                                                       float* __restrict__ y, int rows, int c, int ldx, int ldy,
                                                       int groups, int act, int rows_per_block, int cpg, int ch0) {
  // (`groups` = row length of `stats`; the c channels handled here are channels ch0 .. ch0 + c of the normalised
  // tensor -- x, gamma, beta and y already point at channel ch0 -- so channel k belongs to group (ch0 + k) / cpg)
  const int ch4 = c >> 2;
  const int tpr = ch4 < 256 ? ch4 : 256;   // threads per row
  const int rowlanes = 256 / tpr;
  const int tid = threadIdx.x;
  const int rl = tid / tpr;
  const int cl = tid - rl * tpr;
  if (rl >= rowlanes) return;
  const int n = blockIdx.y;
  const int r0 = blockIdx.x * rows_per_block;
  const int r1 = min(rows, r0 + rows_per_block);
  const float* xb = x + (int64_t)n * rows * ldx;
  float* yb = y + (int64_t)n * rows * ldy;
  const float* st = stats + (int64_t)n * groups * 2;
  for (int c4 = cl; c4 < ch4; c4 += tpr) {
    const float4 g = *reinterpret_cast<const float4*>(gamma + c4 * 4);
    const float4 b = *reinterpret_cast<const float4*>(beta + c4 * 4);
    const float gg[4] = {g.x, g.y, g.z, g.w};
    const float bb[4] = {b.x, b.y, b.z, b.w};
    float mean[4], rstd[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int grp = (ch0 + c4 * 4 + k) / cpg;
      mean[k] = st[grp * 2];
      rstd[k] = st[grp * 2 + 1];
    }
    for (int r = r0 + rl; r < r1; r += 4 * rowlanes) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int rr = r + u * rowlanes;
        if (rr < r1) v[u] = *reinterpret_cast<const float4*>(xb + (int64_t)rr * ldx + c4 * 4);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int rr = r + u * rowlanes;
        if (rr < r1) {
          const float in[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
          float o[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) o[k] = cs_act((in[k] - mean[k]) * rstd[k] * gg[k] + bb[k], act);
          *reinterpret_cast<float4*>(yb + (int64_t)rr * ldy + c4 * 4) = make_float4(o[0], o[1], o[2], o[3]);
        }
      }
    }
  }
}

// GroupNorm apply emitting the fp16 hi / lo pair of y * a_scale (A operand of the F16X3 GEMMs): the fp32 -> pair
// split costs nothing here (this kernel is HBM-bound) and is then done once per element instead of once per
// conv tap inside the GEMM.  Output bytes equal the fp32 version (2 + 2 per element).
typedef _Float16 h4v __attribute__((ext_vector_type(4)));
// Same decomposition as gn_apply_kernel (a workgroup = a run of rows of one sample, a thread keeps its float4 column:
// statistics and affine parameters loaded once, four rows' loads in flight); the first version walked a flat element
// index with two 64-bit divisions per float4 and ran at ~45 % of the HBM rate.
__global__ __launch_bounds__(256) void gn_apply_split16_kernel(const float* __restrict__ x,
                                                               const float* __restrict__ stats,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ beta,
                                                               _Float16* __restrict__ yh, _Float16* __restrict__ yl,
                                                               int rows, int c, int ldx, int ldy, int groups,
                                                               int act, float a_scale, int rows_per_block,
                                                               int32_t* __restrict__ status, int cpg, int ch0) {
  const int ch4 = c >> 2;
  const int tpr = ch4 < 256 ? ch4 : 256;   // threads per row
  const int rowlanes = 256 / tpr;
  const int tid = threadIdx.x;
  const int rl = tid / tpr;
  const int cl = tid - rl * tpr;
  if (rl >= rowlanes) return;
  const int n = blockIdx.y;
  const int r0 = blockIdx.x * rows_per_block;
  const int r1 = min(rows, r0 + rows_per_block);
  const float* xb = x + (int64_t)n * rows * ldx;
  _Float16* hb = yh + (int64_t)n * rows * ldy;
  _Float16* lb = yl + (int64_t)n * rows * ldy;
  const float* st = stats + (int64_t)n * groups * 2;
  float amax = 0.f;
  for (int c4 = cl; c4 < ch4; c4 += tpr) {
    const float4 g = *reinterpret_cast<const float4*>(gamma + c4 * 4);
    const float4 b = *reinterpret_cast<const float4*>(beta + c4 * 4);
    const float gg[4] = {g.x, g.y, g.z, g.w};
    const float bb[4] = {b.x, b.y, b.z, b.w};
    float mean[4], rstd[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int grp = (ch0 + c4 * 4 + k) / cpg;
      mean[k] = st[grp * 2];
      rstd[k] = st[grp * 2 + 1];
    }
    for (int r = r0 + rl; r < r1; r += 4 * rowlanes) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int rr = r + u * rowlanes;
        if (rr < r1) v[u] = *reinterpret_cast<const float4*>(xb + (int64_t)rr * ldx + c4 * 4);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int rr = r + u * rowlanes;
        if (rr < r1) {
          const float in[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
          h4v hi, lo;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float o = cs_act((in[k] - mean[k]) * rstd[k] * gg[k] + bb[k], act) * a_scale;
            amax = fmaxf(amax, fabsf(o));
            const _Float16 h = (_Float16)o;
            hi[k] = h;
            lo[k] = (_Float16)(o - (float)h);
          }
          *reinterpret_cast<h4v*>(hb + (int64_t)rr * ldy + c4 * 4) = hi;
          *reinterpret_cast<h4v*>(lb + (int64_t)rr * ldy + c4 * 4) = lo;
        }
      }
    }
  }
  if (status && amax >= 65504.f) atomicOr(status, CS_STATUS_F16X3_OVERFLOW);
}

// r5: GroupNorm + activation emitted in the WINOGRAD-W form of the 3x3x3 conv that follows (CsConvGemm.a_format = 3): per W
// line of a sample and pair of voxels (w = 2 w2, 2 w2 + 1) the four transformed values [d0 - d2, d1 + d2, d2 - d1, d1 - d3] of
// d_j = y[w - 1 + j] (0 outside the line), split into fp16 hi / lo of value * a_scale -- images [4][nb][lines][W/2][ldv].
// A thread keeps its float4 column and walks whole lines: every activation is evaluated once (a pair's d2, d3 are the
// next pair's d0, d1).
__global__ __launch_bounds__(256) void gn_apply_wino16_kernel(const float* __restrict__ x, const float* __restrict__ stats,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, _Float16* __restrict__ vh,
                                                              _Float16* __restrict__ vl, int lines, int w, int c, int ldx,
                                                              int ldv, int groups, int act, float a_scale,
                                                              int lines_per_block, int64_t pos_stride,
                                                              int32_t* __restrict__ status, int cpg, int ch0) {
  const int ch4 = c >> 2;
  const int tpr = ch4 < 256 ? ch4 : 256;   // threads per line
  const int linelanes = 256 / tpr;
  const int tid = threadIdx.x;
  const int ll = tid / tpr;
  const int cl = tid - ll * tpr;
  if (ll >= linelanes) return;
  const int n = blockIdx.y;
  const int l0 = blockIdx.x * lines_per_block;
  const int l1 = min(lines, l0 + lines_per_block);
  const int w2n = w >> 1;
  const float* xb = x + (int64_t)n * lines * w * ldx;
  const int64_t vrow0 = (int64_t)n * lines * w2n;
  const float* st = stats + (int64_t)n * groups * 2;
  float amax = 0.f;
  for (int c4 = cl; c4 < ch4; c4 += tpr) {
    const float4 g = *reinterpret_cast<const float4*>(gamma + c4 * 4);
    const float4 b = *reinterpret_cast<const float4*>(beta + c4 * 4);
    const float gg[4] = {g.x, g.y, g.z, g.w};
    const float bb[4] = {b.x, b.y, b.z, b.w};
    float mean[4], rstd[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int grp = (ch0 + c4 * 4 + k) / cpg;
      mean[k] = st[grp * 2];
      rstd[k] = st[grp * 2 + 1];
    }
    auto actv = [&](const float4 v, float (&o)[4]) {
      const float in[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = cs_act((in[k] - mean[k]) * rstd[k] * gg[k] + bb[k], act) * a_scale;
    };
    for (int line = l0 + ll; line < l1; line += linelanes) {
      const float* xl = xb + (int64_t)line * w * ldx + c4 * 4;
      float d0[4] = {0.f, 0.f, 0.f, 0.f}, d1[4], d2[4], d3[4];
      actv(*reinterpret_cast<const float4*>(xl), d1);
      for (int w2 = 0; w2 < w2n; ++w2) {
        actv(*reinterpret_cast<const float4*>(xl + (int64_t)(2 * w2 + 1) * ldx), d2);
        if (2 * w2 + 2 < w) {
          actv(*reinterpret_cast<const float4*>(xl + (int64_t)(2 * w2 + 2) * ldx), d3);
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k) d3[k] = 0.f;
        }
        const int64_t off = (vrow0 + (int64_t)line * w2n + w2) * ldv + c4 * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          h4v hi, lo;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float o = q == 0 ? d0[k] - d2[k] : q == 1 ? d1[k] + d2[k] : q == 2 ? d2[k] - d1[k] : d1[k] - d3[k];
            amax = fmaxf(amax, fabsf(o));
            const _Float16 hh = (_Float16)o;
            hi[k] = hh;
            lo[k] = (_Float16)(o - (float)hh);
          }
          *reinterpret_cast<h4v*>(vh + q * pos_stride + off) = hi;
          *reinterpret_cast<h4v*>(vl + q * pos_stride + off) = lo;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          d0[k] = d2[k];
          d1[k] = d3[k];
        }
      }
    }
  }
  if (status && amax >= 65504.f) atomicOr(status, CS_STATUS_F16X3_OVERFLOW);
}

// F(4,3) along W (CsConvGemm.a_format = 4): per W line and FOUR voxels (w = 4 t .. 4 t + 3) the six transformed values B^T d of
// d_j = y[4 t - 1 + j] (0 outside the line) -- images [6][nb][lines][W/4][ldv].  Same line walk: a tile's d4, d5 are the next
// tile's d0, d1.
// (V = channels per thread.  r6 measured V = 8 -- two float4 loads, ONE 16-byte store per image and position instead of two
// 8-byte ones -- and it LOST: 80.7 vs 61.3 us per launch, 64.63 vs 64.00 ms per 32-object step same box
// (profiles/r06_j_gn_wino_v8_whatif_ab.txt): half the threads in flight cost more than the wider stores gain.  V = 4 stays;
// -DCS_GN_WINO_V8 is the what-if build.  Same arithmetic per element either way.)
template <int V>
__global__ __launch_bounds__(256) void gn_apply_wino43_kernel(const float* __restrict__ x, const float* __restrict__ stats,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, _Float16* __restrict__ vh,
                                                              _Float16* __restrict__ vl, int lines, int w, int c, int ldx,
                                                              int ldv, int groups, int act, float a_scale,
                                                              int lines_per_block, int64_t pos_stride,
                                                              int32_t* __restrict__ status, int cpg, int ch0) {
  typedef _Float16 hv __attribute__((ext_vector_type(V)));
  const int chv = c / V;
  const int tpr = chv < 256 ? chv : 256;
  const int linelanes = 256 / tpr;
  const int tid = threadIdx.x;
  const int ll = tid / tpr;
  const int cl = tid - ll * tpr;
  if (ll >= linelanes) return;
  const int n = blockIdx.y;
  const int l0 = blockIdx.x * lines_per_block;
  const int l1 = min(lines, l0 + lines_per_block);
  const int w4n = w >> 2;
  const float* xb = x + (int64_t)n * lines * w * ldx;
  const int64_t vrow0 = (int64_t)n * lines * w4n;
  const float* st = stats + (int64_t)n * groups * 2;
  float amax = 0.f;
  for (int cv = cl; cv < chv; cv += tpr) {
    float gg[V], bb[V], mean[V], rstd[V];
#pragma unroll
    for (int k4 = 0; k4 < V / 4; ++k4) {
      const float4 g = *reinterpret_cast<const float4*>(gamma + cv * V + 4 * k4);
      const float4 b = *reinterpret_cast<const float4*>(beta + cv * V + 4 * k4);
      gg[4 * k4] = g.x; gg[4 * k4 + 1] = g.y; gg[4 * k4 + 2] = g.z; gg[4 * k4 + 3] = g.w;
      bb[4 * k4] = b.x; bb[4 * k4 + 1] = b.y; bb[4 * k4 + 2] = b.z; bb[4 * k4 + 3] = b.w;
    }
#pragma unroll
    for (int k = 0; k < V; ++k) {
      const int grp = (ch0 + cv * V + k) / cpg;
      mean[k] = st[grp * 2];
      rstd[k] = st[grp * 2 + 1];
    }
    auto actv = [&](const float* px, bool in, float (&o)[V]) {
      if (!in) {
#pragma unroll
        for (int k = 0; k < V; ++k) o[k] = 0.f;
        return;
      }
#pragma unroll
      for (int k4 = 0; k4 < V / 4; ++k4) {
        const float4 v = *reinterpret_cast<const float4*>(px + 4 * k4);
        const float iv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k)
          o[4 * k4 + k] = cs_act((iv[k] - mean[4 * k4 + k]) * rstd[4 * k4 + k] * gg[4 * k4 + k] + bb[4 * k4 + k], act) * a_scale;
      }
    };
    for (int line = l0 + ll; line < l1; line += linelanes) {
      const float* xl = xb + (int64_t)line * w * ldx + cv * V;
      float d[6][V];
#pragma unroll
      for (int k = 0; k < V; ++k) d[0][k] = 0.f;
      actv(xl, true, d[1]);
      for (int t = 0; t < w4n; ++t) {
#pragma unroll
        for (int j = 2; j < 6; ++j) {
          const int wi = 4 * t - 1 + j;
          actv(xl + (int64_t)wi * ldx, wi < w, d[j]);
        }
        const int64_t off = (vrow0 + (int64_t)line * w4n + t) * ldv + cv * V;
#pragma unroll
        for (int q = 0; q < 6; ++q) {
          hv hi, lo;
#pragma unroll
          for (int k = 0; k < V; ++k) {
            float o;
            if (q == 0) o = 4.f * d[0][k] - 5.f * d[2][k] + d[4][k];
            else if (q == 1) o = (d[4][k] + d[3][k]) - 4.f * (d[1][k] + d[2][k]);
            else if (q == 2) o = 4.f * (d[1][k] - d[2][k]) + (d[4][k] - d[3][k]);
            else if (q == 3) o = 2.f * (d[3][k] - d[1][k]) + (d[4][k] - d[2][k]);
            else if (q == 4) o = 2.f * (d[1][k] - d[3][k]) + (d[4][k] - d[2][k]);
            else o = 4.f * d[1][k] - 5.f * d[3][k] + d[5][k];
            amax = fmaxf(amax, fabsf(o));
            const _Float16 hh = (_Float16)o;
            hi[k] = hh;
            lo[k] = (_Float16)(o - (float)hh);
          }
          *reinterpret_cast<hv*>(vh + q * pos_stride + off) = hi;
          *reinterpret_cast<hv*>(vl + q * pos_stride + off) = lo;
        }
#pragma unroll
        for (int k = 0; k < V; ++k) {
          d[0][k] = d[4][k];
          d[1][k] = d[5][k];
        }
      }
    }
  }
  if (status && amax >= 65504.f) atomicOr(status, CS_STATUS_F16X3_OVERFLOW);
}

// LayerNorm: one wave per row; each lane owns up to MAXV float4 chunks (c <= 64*4*MAXV).
template <int MAXV>
__global__ __launch_bounds__(256) void ln_kernel(const float* __restrict__ x,
                                                 const float* __restrict__ gamma,
                                                 const float* __restrict__ beta,
                                                 float* __restrict__ y, int m, int c, int ldx,
                                                 int ldy, float eps) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int ch4 = c >> 2;
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < m; row += (int64_t)gridDim.x * 4) {
    float4 v[MAXV];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
      const int c4 = lane + 64 * k;
      v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c4 < ch4) v[k] = *reinterpret_cast<const float4*>(x + row * ldx + c4 * 4);
      s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
    }
    const float mean = wave_sum(s) / (float)c;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
      const int c4 = lane + 64 * k;
      if (c4 < ch4) {
        const float a = v[k].x - mean, b = v[k].y - mean, cc = v[k].z - mean, d = v[k].w - mean;
        q += (a * a + b * b) + (cc * cc + d * d);
      }
    }
    const float var = wave_sum(q) / (float)c;
    const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
      const int c4 = lane + 64 * k;
      if (c4 < ch4) {
        const float4 g = *reinterpret_cast<const float4*>(gamma + c4 * 4);
        const float4 b = *reinterpret_cast<const float4*>(beta + c4 * 4);
        float4 o;
        o.x = (v[k].x - mean) * rstd * g.x + b.x;
        o.y = (v[k].y - mean) * rstd * g.y + b.y;
        o.z = (v[k].z - mean) * rstd * g.z + b.z;
        o.w = (v[k].w - mean) * rstd * g.w + b.w;
        *reinterpret_cast<float4*>(y + row * ldy + c4 * 4) = o;
      }
    }
  }
}

// LayerNorm emitting the interleaved F16X3 operand pair (CsConvGemm.a_format = 2): per row and 16-channel chunk the 64
// bytes [hi c0-7 | lo c0-7 | hi c8-15 | lo c8-15], halves of y * a_scale -- same bytes and row stride as the fp32 output
// it replaces, so the consuming GEMM gathers the same 64-byte pieces and its K loop carries no conversion.
template <int MAXV>
__global__ __launch_bounds__(256) void ln_pair_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, _Float16* __restrict__ y, int m,
                                                      int c, int ldx, int ldy, float eps, float a_scale,
                                                      int32_t* __restrict__ status) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int ch4 = c >> 2;
  float amax = 0.f;
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < m; row += (int64_t)gridDim.x * 4) {
    float4 v[MAXV];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
      const int c4 = lane + 64 * k;
      v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c4 < ch4) v[k] = *reinterpret_cast<const float4*>(x + row * ldx + c4 * 4);
      s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
    }
    const float mean = wave_sum(s) / (float)c;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
      const int c4 = lane + 64 * k;
      if (c4 < ch4) {
        const float a = v[k].x - mean, b = v[k].y - mean, cc = v[k].z - mean, d = v[k].w - mean;
        q += (a * a + b * b) + (cc * cc + d * d);
      }
    }
    const float var = wave_sum(q) / (float)c;
    const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
      const int c4 = lane + 64 * k;
      if (c4 < ch4) {
        const float4 g = *reinterpret_cast<const float4*>(gamma + c4 * 4);
        const float4 b = *reinterpret_cast<const float4*>(beta + c4 * 4);
        // the fp32 LayerNorm's expression, then the operand scale (a power of two: exact)
        const float o[4] = {((v[k].x - mean) * rstd * g.x + b.x) * a_scale, ((v[k].y - mean) * rstd * g.y + b.y) * a_scale,
                            ((v[k].z - mean) * rstd * g.z + b.z) * a_scale, ((v[k].w - mean) * rstd * g.w + b.w) * a_scale};
        h4v hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          amax = fmaxf(amax, fabsf(o[e]));
          const _Float16 h = (_Float16)o[e];
          hi[e] = h;
          lo[e] = (_Float16)(o[e] - (float)h);
        }
        // channel 4*c4 + e sits in chunk (4*c4) / 16 at j = (4*c4) % 16: hi at halves (j < 8 ? 0 : 16) + j % 8, lo 8 further
        const int cch = c4 >> 2, j = (c4 & 3) * 4;
        _Float16* dst = y + row * (int64_t)ldy * 2 + cch * 32 + (j < 8 ? 0 : 16) + (j & 7);
        *reinterpret_cast<h4v*>(dst) = hi;
        *reinterpret_cast<h4v*>(dst + 8) = lo;
      }
    }
  }
  if (status && amax >= 65504.f) atomicOr(status, CS_STATUS_F16X3_OVERFLOW);
}

// Small tensors (one or two objects: a few MB, L2-resident): statistics and normalisation in ONE launch, one workgroup
// per (sample, group).  The three-launch path costs ~23 us per GroupNorm there (6.8 + 4.6 + 11.5 us, each at its launch
// floor), 61 times a step.  Threads walk the group's rows x cpg elements in a fixed stride, accumulate in fp64, and a
// fixed LDS tree adds the 256 partials, so the statistics are reproducible; the second sweep re-reads the group from
// L2 and applies the same expression as gn_apply_kernel.
__global__ __launch_bounds__(256) void gn_small_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float* __restrict__ y,
                                                       float* __restrict__ stats, int rows, int c, int ldx, int ldy,
                                                       int groups, float eps, int act) {
  __shared__ double red[2][256];
  const int n = blockIdx.x / groups, g = blockIdx.x - n * groups;
  const int cpg = c / groups;
  const int tid = threadIdx.x;
  const float* xb = x + (int64_t)n * rows * ldx + g * cpg;
  float* yb = y + (int64_t)n * rows * ldy + g * cpg;
  const int dr = 256 / cpg, dk = 256 - dr * cpg;          // element index + 256  ->  (row + dr, k + dk) with carry
  auto next = [&](int& r, int& k) {
    r += dr;
    k += dk;
    if (k >= cpg) {
      k -= cpg;
      ++r;
    }
  };
  double s = 0, q = 0;
  {
    int r = tid / cpg, k = tid - r * cpg;
    while (r < rows) {                                      // four loads in flight, accumulated in element order
      float v[4];
      int rr[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        rr[u] = r;
        v[u] = r < rows ? xb[(int64_t)r * ldx + k] : 0.f;
        next(r, k);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (rr[u] < rows) {
          s += v[u];
          q += (double)v[u] * v[u];
        }
    }
  }
  red[0][tid] = s;
  red[1][tid] = q;
  __syncthreads();
#pragma unroll
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) {
      red[0][tid] += red[0][tid + o];
      red[1][tid] += red[1][tid + o];
    }
    __syncthreads();
  }
  const double count = (double)rows * cpg;
  const double mean_d = red[0][0] / count;
  double var = red[1][0] / count - mean_d * mean_d;
  if (var < 0) var = 0;
  const float mean = (float)mean_d;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  if (tid == 0 && stats) {
    stats[2 * blockIdx.x] = mean;
    stats[2 * blockIdx.x + 1] = rstd;
  }
  {
    int r = tid / cpg, k = tid - r * cpg;
    while (r < rows) {
      float v[4], ga[4], be[4];
      int rr[4], kk[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        rr[u] = r;
        kk[u] = k;
        if (r < rows) {
          v[u] = xb[(int64_t)r * ldx + k];
          ga[u] = gamma[g * cpg + k];
          be[u] = beta[g * cpg + k];
        }
        next(r, k);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (rr[u] < rows) yb[(int64_t)rr[u] * ldy + kk[u]] = cs_act((v[u] - mean) * rstd * ga[u] + be[u], act);
    }
  }
}

// gn_small_kernel with the statistics taken from the producers' partials (r4): one workgroup per (sample, group) adds the
// group's partial sums (256 threads over the (channel, tile) pairs, fixed LDS tree) and makes ONE sweep over the group.
// Small tensors only (the gn_small rule): one launch per GroupNorm where the two-launch form would sit at its launch floors.
__global__ __launch_bounds__(256) void gn_small_parts_kernel(const float* __restrict__ x, const GnSegs sg,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float* __restrict__ y,
                                                             float* __restrict__ stats, float* __restrict__ bound, int rows,
                                                             int c, int ldx, int ldy, int groups, float eps, int act,
                                                             int rsplit) {
  // r5: `rsplit` workgroups per (sample, group), each deriving the SAME statistics from the partials (a few hundred pairs)
  // and sweeping its share of the rows -- one or two objects are 64 (sample, group)s, a quarter of the CUs
  __shared__ double red[2][256];
  const int sgi = blockIdx.x / rsplit, rs = blockIdx.x - sgi * rsplit;
  const int n = sgi / groups, g = sgi - n * groups;
  const int cpg = c / groups;
  const int tid = threadIdx.x;
  double s, q;
  gn_group_parts(sg, n, g, cpg, tid, 256, s, q);
  red[0][tid] = s;
  red[1][tid] = q;
  __syncthreads();
#pragma unroll
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) {
      red[0][tid] += red[0][tid + o];
      red[1][tid] += red[1][tid + o];
    }
    __syncthreads();
  }
  const double count = (double)rows * cpg;
  const double mean_d = red[0][0] / count;
  double var = red[1][0] / count - mean_d * mean_d;
  if (var < 0) var = 0;
  const float mean = (float)mean_d;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  if (tid == 0 && stats && rs == 0) {
    stats[2 * sgi] = mean;
    stats[2 * sgi + 1] = rstd;
  }
  if (tid == 0 && bound && rs == 0) gn_bound_max(bound, mean_d, var, count);
  const int rper = (rows + rsplit - 1) / rsplit;
  const int rbeg = rs * rper, rend = min(rows, rbeg + rper);
  const float* xb = x + (int64_t)n * rows * ldx + g * cpg;
  float* yb = y + (int64_t)n * rows * ldy + g * cpg;
  const int dr = 256 / cpg, dk = 256 - dr * cpg;
  int r = rbeg + tid / cpg, k = tid - (tid / cpg) * cpg;
  rows = rend;                                    // (this workgroup's share ends here; xb / yb were formed above)
  while (r < rows) {
    float v[4], ga[4], be[4];
    int rr[4], kk[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      rr[u] = r;
      kk[u] = k;
      if (r < rows) {
        v[u] = xb[(int64_t)r * ldx + k];
        ga[u] = gamma[g * cpg + k];
        be[u] = beta[g * cpg + k];
      }
      r += dr;
      k += dk;
      if (k >= cpg) {
        k -= cpg;
        ++r;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (rr[u] < rows) yb[(int64_t)rr[u] * ldy + kk[u]] = cs_act((v[u] - mean) * rstd * ga[u] + be[u], act);
  }
}

// row slices per sample: enough workgroups to fill the chip (~2048 in total) but no slice shorter than
// GN_MIN_SPLIT_ROWS rows (each workgroup ends in a serial LDS reduction that longer slices amortise)
int gn_nsplit(int rows, int nb) {
  int n = (2048 + nb - 1) / nb;
  const int cap = (rows + GN_MIN_SPLIT_ROWS - 1) / GN_MIN_SPLIT_ROWS;
  if (n > cap) n = cap;
  if (n < 1) n = 1;
  return n < GN_MAX_SPLITS ? n : GN_MAX_SPLITS;
}

}  // namespace

extern "C" int64_t cs_groupnorm_ws_bytes(int nb, int groups) {
  return (int64_t)nb * GN_MAX_SPLITS * groups * 2 * (int64_t)sizeof(double);
}

static int gn_pack_segs(const CsGnSeg* segs, int nseg, int nb, int c, GnSegs& sg) {
  if (!segs || nseg < 1 || nseg > 4) return CS_EINVAL;
  sg.n = nseg;
  int next = 0;
  for (int i = 0; i < nseg; ++i) {
    const CsGnSeg& s = segs[i];
    // contiguous cover of [0, c) in channel order; every tile index the kernel forms must exist
    if (!s.part || ((uintptr_t)s.part & 15) || s.ch0 != next || s.nch <= 0 || s.col0 < 0 || s.col0 + s.nch > s.ld ||
        s.tiles_per_sample <= 0 || s.ncls <= 0 || s.nb_src <= 0 || (nb % s.nb_src) != 0)
      return CS_EINVAL;
    next += s.nch;
    sg.s[i] = s;
  }
  if (next != c) return CS_EINVAL;
  for (int i = nseg; i < 4; ++i) sg.s[i] = segs[0];
  return CS_OK;
}

extern "C" int cs_groupnorm_finalize_parts(const CsGnSeg* segs, int nseg, int nb, int rows, int c, int groups, float eps,
                                           float* stats, float* bound, cs_stream_t stream) {
  if ((!stats && !bound) || ((uintptr_t)bound & 3) || nb <= 0 || rows <= 0 || c <= 0 || groups <= 0 || c % groups) return CS_EINVAL;
  GnSegs sg;
  const int rc = gn_pack_segs(segs, nseg, nb, c, sg);
  if (rc != CS_OK) return rc;
  const int total = nb * groups;
  // (channel, tile) pairs of the largest group share: four waves per group from 512 pairs (see the kernel)
  int64_t pairs = 0;
  for (int i = 0; i < nseg; ++i) {
    const int64_t nt = (int64_t)segs[i].ncls * segs[i].tiles_per_sample;
    const int64_t chs = segs[i].nch < c / groups ? segs[i].nch : c / groups;
    if (nt * chs > pairs) pairs = nt * chs;
  }
  if (pairs >= 512 && !cs_debug()->no_gn_fold) {
    CS_LAUNCH(gn_finalize_parts_kernel<4>, dim3((total + 3) / 4), dim3(64 * GN_FIN_WAVES), 0, (hipStream_t)stream, sg, groups,
              c / groups, (double)rows * (c / groups), eps, stats, bound, total);
  } else {
    CS_LAUNCH(gn_finalize_parts_kernel<1>, dim3((total + GN_FIN_WAVES - 1) / GN_FIN_WAVES), dim3(64 * GN_FIN_WAVES), 0,
              (hipStream_t)stream, sg, groups, c / groups, (double)rows * (c / groups), eps, stats, bound, total);
  }
  CS_CHECK_LAUNCH();
  return CS_OK;
}

// GroupNorm of a tensor whose statistics come from its producers' partials, in ONE call: a single launch for small
// tensors (the cs_groupnorm rule: <= 16 MB, <= 11264 elements per (sample, group)), cs_groupnorm_finalize_parts +
// cs_groupnorm_apply otherwise.  `stats` [nb][groups][2] is written either way.
extern "C" int cs_groupnorm_parts(const float* x, const CsGnSeg* segs, int nseg, const float* gamma, const float* beta,
                                  float* y, int nb, int rows, int c, int ldx, int ldy, int groups, float eps, int act,
                                  float* stats, float* bound, cs_stream_t stream) {
  if (!x || !gamma || !beta || !y || !stats || ((uintptr_t)bound & 3) || nb <= 0 || rows <= 0 || c <= 0 || groups <= 0 || c % groups || ldx < c ||
      ldy < c)
    return CS_EINVAL;
  GnSegs sg;
  const int rc0 = gn_pack_segs(segs, nseg, nb, c, sg);
  if (rc0 != CS_OK) return rc0;
  const int cpg = c / groups;
  const int64_t small_group = cs_debug()->gn_small_group;
  if ((int64_t)nb * rows * c * 4 <= GN_SMALL_BYTES && cpg <= 256 && (int64_t)rows * cpg <= small_group &&
      (int64_t)nb * groups <= 65535) {
    // (row shares per (sample, group): up to four while the launch stays under ~512 workgroups and a share keeps >= 4
    // elements per thread; depends on the sample's own shape and on nb only through the workgroup budget -- the
    // arithmetic per element does not depend on it)
    int rsplit = 1;
    while (rsplit < 4 && (int64_t)nb * groups * rsplit * 2 <= 512 && (int64_t)rows * cpg >= 2048LL * rsplit) rsplit *= 2;
    if (cs_debug()->no_gn_fold) rsplit = 1;
    CS_LAUNCH(gn_small_parts_kernel, dim3((unsigned)(nb * groups * rsplit)), dim3(256), 0, (hipStream_t)stream, x, sg, gamma,
              beta, y, stats, bound, rows, c, ldx, ldy, groups, eps, act, rsplit);
    CS_CHECK_LAUNCH();
    return CS_OK;
  }
  const int rc = cs_groupnorm_finalize_parts(segs, nseg, nb, rows, c, groups, eps, stats, bound, stream);
  if (rc) return rc;
  return cs_groupnorm_apply(x, stats, gamma, beta, y, nb, rows, c, ldx, ldy, groups, act, stream);
}

static int gn_stats_impl(const float* x, int nb, int rows, int c, int ldx, int groups, float eps, void* ws, float* stats,
                         float* bound, cs_stream_t stream);

extern "C" int cs_groupnorm_stats(const float* x, int nb, int rows, int c, int ldx, int groups,
                                  float eps, void* ws, float* stats, cs_stream_t stream) {
  return gn_stats_impl(x, nb, rows, c, ldx, groups, eps, ws, stats, nullptr, stream);
}

// cs_groupnorm_stats that also leaves the tensor's magnitude bound (max over (sample, group) of |mean| + std sqrt(n - 1),
// see cs_groupnorm_finalize_parts) in `bound`: for a tensor whose producers left no partial sums (r4)
extern "C" int cs_groupnorm_stats_bound(const float* x, int nb, int rows, int c, int ldx, int groups, float eps, void* ws,
                                        float* stats, float* bound, cs_stream_t stream) {
  if (!bound || ((uintptr_t)bound & 3)) return CS_EINVAL;
  return gn_stats_impl(x, nb, rows, c, ldx, groups, eps, ws, stats, bound, stream);
}

static int gn_stats_impl(const float* x, int nb, int rows, int c, int ldx, int groups, float eps, void* ws, float* stats,
                         float* bound, cs_stream_t stream) {
  if (!x || !ws || !stats || nb <= 0 || rows <= 0 || c <= 0 || groups <= 0) return CS_EINVAL;
  if ((c & 3) || (ldx & 3) || ldx < c || c % groups || groups > 256) return CS_EINVAL;
  if (((uintptr_t)x & 15) || ((uintptr_t)ws & 7)) return CS_EINVAL;
  const int nsplit = gn_nsplit(rows, nb);
  const int rps = (rows + nsplit - 1) / nsplit;
  const int ch4 = c >> 2;
  const int tpr = ch4 < 256 ? ch4 : 256;
  const int rowlanes = 256 / tpr;
  const size_t smem = (size_t)rowlanes * c * 2 * sizeof(double);
  if (smem > 64 * 1024) return CS_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  CS_LAUNCH(gn_partial_kernel, dim3((unsigned)(nb * nsplit)), dim3(256), smem, s, x, rows,
                     c, ldx, groups, nsplit, rps, (double*)ws);
  CS_CHECK_LAUNCH();
  const int total = nb * groups;
  CS_LAUNCH(gn_finalize_kernel, dim3((total + 3) / 4), dim3(256), 0, s,
                     (const double*)ws, nsplit, groups, (double)rows * (c / groups), eps, stats,
                     total, bound);
  CS_CHECK_LAUNCH();
  return CS_OK;
}

// Channel-range form of the two apply entries: the c channels handled are channels ch0 .. ch0 + c of a tensor that was
// normalised over `groups` groups of `cpg` channels each (stats: [nb][groups][2]); x, gamma, beta and y point AT channel
// ch0.  Lets one statistics pass over a concatenated tensor [h | skip] feed two separate operand tensors (the channel
// split of the classifier-free-guidance output blocks, unet.py::_res_split).
extern "C" int cs_groupnorm_apply_range(const float* x, const float* stats, const float* gamma, const float* beta,
                                        float* y, int nb, int rows, int c, int ldx, int ldy, int groups, int cpg,
                                        int ch0, int act, cs_stream_t stream) {
  if (!x || !stats || !gamma || !beta || !y || nb <= 0 || rows <= 0 || c <= 0 || groups <= 0 || cpg <= 0 || ch0 < 0)
    return CS_EINVAL;
  if ((c & 3) || (ldx & 3) || (ldy & 3) || ldx < c || ldy < c || (int64_t)ch0 + c > (int64_t)groups * cpg) return CS_EINVAL;
  if (((uintptr_t)x & 15) || ((uintptr_t)y & 15) || ((uintptr_t)gamma & 15) ||
      ((uintptr_t)beta & 15))
    return CS_EINVAL;
  if (nb > 65535) return CS_EINVAL;
  // enough workgroups to fill the chip (>= ~2048 in total), each with at least 16 rows per row-lane
  const int ch4 = c >> 2;
  const int rowlanes = 256 / (ch4 < 256 ? ch4 : 256);
  int blocks_per_sample = (2048 + nb - 1) / nb;
  // (r5: down to four rows per row-lane -- one round of the kernel's four loads in flight -- where sixteen would leave
  // most of the chip idle: 8192 x 224 at one object ran 128 workgroups x 4 dependent rounds, 12 us for 14 MB)
  const int min_rows = (int64_t)nb * ((rows + 16 * rowlanes - 1) / (16 * rowlanes)) >= 1024 ? 16 : 4;
  const int max_blocks = (rows + min_rows * rowlanes - 1) / (min_rows * rowlanes);
  if (blocks_per_sample > max_blocks) blocks_per_sample = max_blocks;
  if (blocks_per_sample < 1) blocks_per_sample = 1;
  const int rpb = (rows + blocks_per_sample - 1) / blocks_per_sample;
  CS_LAUNCH(gn_apply_kernel, dim3((unsigned)((rows + rpb - 1) / rpb), (unsigned)nb), dim3(256), 0,
            (hipStream_t)stream, x, stats, gamma, beta, y, rows, c, ldx, ldy, groups, act, rpb, cpg, ch0);
  CS_CHECK_LAUNCH();
  return CS_OK;
}

extern "C" int cs_groupnorm_apply(const float* x, const float* stats, const float* gamma,
                                  const float* beta, float* y, int nb, int rows, int c, int ldx,
                                  int ldy, int groups, int act, cs_stream_t stream) {
  if (groups <= 0 || c <= 0 || c % groups) return CS_EINVAL;
  return cs_groupnorm_apply_range(x, stats, gamma, beta, y, nb, rows, c, ldx, ldy, groups, c / groups, 0, act, stream);
}

extern "C" int cs_groupnorm_apply_split16_range(const float* x, const float* stats, const float* gamma,
                                                const float* beta, void* y_hi, void* y_lo, int nb, int rows, int c,
                                                int ldx, int ldy, int groups, int cpg, int ch0, int act, float a_scale,
                                                int32_t* status, cs_stream_t stream) {
  if (!x || !stats || !gamma || !beta || !y_hi || !y_lo || nb <= 0 || rows <= 0 || c <= 0 || groups <= 0 || cpg <= 0 ||
      ch0 < 0)
    return CS_EINVAL;
  if ((c & 7) || (ldx & 3) || (ldy & 7) || ldx < c || ldy < c || (int64_t)ch0 + c > (int64_t)groups * cpg ||
      !(a_scale > 0.f))
    return CS_EINVAL;
  if (((uintptr_t)x & 15) || ((uintptr_t)y_hi & 15) || ((uintptr_t)y_lo & 15) || ((uintptr_t)gamma & 15) ||
      ((uintptr_t)beta & 15))
    return CS_EINVAL;
  if (nb > 65535) return CS_EINVAL;
  const int ch4 = c >> 2;
  const int rowlanes = 256 / (ch4 < 256 ? ch4 : 256);
  int blocks_per_sample = (2048 + nb - 1) / nb;
  // (r5: down to four rows per row-lane -- one round of the kernel's four loads in flight -- where sixteen would leave
  // most of the chip idle: 8192 x 224 at one object ran 128 workgroups x 4 dependent rounds, 12 us for 14 MB)
  const int min_rows = (int64_t)nb * ((rows + 16 * rowlanes - 1) / (16 * rowlanes)) >= 1024 ? 16 : 4;
  const int max_blocks = (rows + min_rows * rowlanes - 1) / (min_rows * rowlanes);
  if (blocks_per_sample > max_blocks) blocks_per_sample = max_blocks;
  if (blocks_per_sample < 1) blocks_per_sample = 1;
  const int rpb = (rows + blocks_per_sample - 1) / blocks_per_sample;
  CS_LAUNCH(gn_apply_split16_kernel, dim3((unsigned)((rows + rpb - 1) / rpb), (unsigned)nb), dim3(256), 0,
            (hipStream_t)stream, x, stats, gamma, beta, (_Float16*)y_hi, (_Float16*)y_lo, rows, c, ldx, ldy, groups, act,
            a_scale, rpb, status, cpg, ch0);
  CS_CHECK_LAUNCH();
  return CS_OK;
}

extern "C" int cs_groupnorm_apply_wino_range(const float* x, const float* stats, const float* gamma, const float* beta,
                                             void* v_hi, void* v_lo, int nb, int d, int h, int w, int c, int ldx, int ldv,
                                             int groups, int cpg, int ch0, int act, float a_scale, int variant,
                                             int32_t* status, cs_stream_t stream) {
  if (!x || !stats || !gamma || !beta || !v_hi || !v_lo || nb <= 0 || d <= 0 || h <= 0 || w < 2 || c <= 0 || groups <= 0 ||
      cpg <= 0 || ch0 < 0 || (variant != 2 && variant != 4))
    return CS_EINVAL;
  if ((w % variant) || (c & 7) || (int64_t)ch0 + c > (int64_t)groups * cpg || (ldx & 3) || (ldv & 7) || ldx < c || ldv < c ||
      !(a_scale > 0.f) || nb > 65535)
    return CS_EINVAL;
  if (((uintptr_t)x & 15) || ((uintptr_t)v_hi & 15) || ((uintptr_t)v_lo & 15) || ((uintptr_t)gamma & 15) ||
      ((uintptr_t)beta & 15))
    return CS_EINVAL;
  const int64_t lines64 = (int64_t)d * h;
  if (lines64 * w > 0x7fffffffLL) return CS_EINVAL;
  const int lines = (int)lines64;
#ifdef CS_GN_WINO_V8           // what-if build: eight channels per thread (one 16-byte store per image and position)
  constexpr int WV = 8;
#else
  constexpr int WV = 4;            // channels per thread of the F(4,3) kernel
#endif
  const int ch4 = variant == 4 ? c / WV : c >> 2;
  const int linelanes = 256 / (ch4 < 256 ? ch4 : 256);
  // ~2048 workgroups in all, at least one line per line-lane
  int blocks_per_sample = (2048 + nb - 1) / nb;
  const int max_blocks = (lines + linelanes - 1) / linelanes;
  if (blocks_per_sample > max_blocks) blocks_per_sample = max_blocks;
  if (blocks_per_sample < 1) blocks_per_sample = 1;
  const int lpb = (lines + blocks_per_sample - 1) / blocks_per_sample;
  const int64_t pos_stride = (int64_t)nb * lines * (w / variant) * ldv;
  if (variant == 4) {
    CS_LAUNCH(gn_apply_wino43_kernel<WV>, dim3((unsigned)((lines + lpb - 1) / lpb), (unsigned)nb), dim3(256), 0,
              (hipStream_t)stream, x, stats, gamma, beta, (_Float16*)v_hi, (_Float16*)v_lo, lines, w, c, ldx, ldv, groups, act,
              a_scale, lpb, pos_stride, status, cpg, ch0);
  } else {
    CS_LAUNCH(gn_apply_wino16_kernel, dim3((unsigned)((lines + lpb - 1) / lpb), (unsigned)nb), dim3(256), 0,
              (hipStream_t)stream, x, stats, gamma, beta, (_Float16*)v_hi, (_Float16*)v_lo, lines, w, c, ldx, ldv, groups, act,
              a_scale, lpb, pos_stride, status, cpg, ch0);
  }
  CS_CHECK_LAUNCH();
  return CS_OK;
}

extern "C" int cs_groupnorm_apply_wino16_range(const float* x, const float* stats, const float* gamma, const float* beta,
                                               void* v_hi, void* v_lo, int nb, int d, int h, int w, int c, int ldx, int ldv,
                                               int groups, int cpg, int ch0, int act, float a_scale, int32_t* status,
                                               cs_stream_t stream) {
  return cs_groupnorm_apply_wino_range(x, stats, gamma, beta, v_hi, v_lo, nb, d, h, w, c, ldx, ldv, groups, cpg, ch0, act,
                                       a_scale, 2, status, stream);
}

extern "C" int cs_groupnorm_apply_wino16(const float* x, const float* stats, const float* gamma, const float* beta,
                                         void* v_hi, void* v_lo, int nb, int d, int h, int w, int c, int ldx, int ldv,
                                         int groups, int act, float a_scale, int32_t* status, cs_stream_t stream) {
  if (groups <= 0 || c <= 0 || c % groups) return CS_EINVAL;
  return cs_groupnorm_apply_wino16_range(x, stats, gamma, beta, v_hi, v_lo, nb, d, h, w, c, ldx, ldv, groups, c / groups, 0,
                                         act, a_scale, status, stream);
}

extern "C" int cs_groupnorm_apply_split16(const float* x, const float* stats, const float* gamma,
                                          const float* beta, void* y_hi, void* y_lo, int nb, int rows, int c,
                                          int ldx, int ldy, int groups, int act, float a_scale,
                                          int32_t* status, cs_stream_t stream) {
  if (groups <= 0 || c <= 0 || c % groups) return CS_EINVAL;
  return cs_groupnorm_apply_split16_range(x, stats, gamma, beta, y_hi, y_lo, nb, rows, c, ldx, ldy, groups, c / groups, 0,
                                          act, a_scale, status, stream);
}

extern "C" int cs_groupnorm(const float* x, const float* gamma, const float* beta, float* y, int nb, int rows, int c,
                            int ldx, int ldy, int groups, float eps, int act, void* ws, float* stats,
                            cs_stream_t stream) {
  if (!x || !gamma || !beta || !y || !stats || nb <= 0 || rows <= 0 || c <= 0 || groups <= 0) return CS_EINVAL;
  if (c % groups || ldx < c || ldy < c) return CS_EINVAL;
  const int cpg = c / groups;
  const int64_t small_group = cs_debug()->gn_small_group;      // (CS_GN_SMALL_GROUP: tuning override, tools/gn_bench.py)
  // one launch while the tensor is a few MB (it stays in L2 between the two sweeps) and a group is a handful of
  // elements per thread; otherwise statistics (two launches) + apply
  if ((int64_t)nb * rows * c * 4 <= GN_SMALL_BYTES && cpg <= 256 && (int64_t)rows * cpg <= small_group &&
      (int64_t)nb * groups <= 65535) {
    CS_LAUNCH(gn_small_kernel, dim3((unsigned)(nb * groups)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y,
              stats, rows, c, ldx, ldy, groups, eps, act);
    CS_CHECK_LAUNCH();
    return CS_OK;
  }
  if (!ws) return CS_EINVAL;
  int rc = cs_groupnorm_stats(x, nb, rows, c, ldx, groups, eps, ws, stats, stream);
  if (rc) return rc;
  return cs_groupnorm_apply(x, stats, gamma, beta, y, nb, rows, c, ldx, ldy, groups, act, stream);
}

extern "C" int cs_groupnorm_silu_ndhwc(const float* x, const float* gamma, const float* beta,
                                       float* y, int nb, int rows, int c, int groups, float eps,
                                       void* ws, float* stats, cs_stream_t stream) {
  return cs_groupnorm(x, gamma, beta, y, nb, rows, c, c, c, groups, eps, CS_ACT_SILU, ws, stats, stream);
}

extern "C" int cs_layernorm(const float* x, const float* gamma, const float* beta, float* y, int m,
                            int c, int ldx, int ldy, float eps, cs_stream_t stream) {
  if (!x || !gamma || !beta || !y || m <= 0 || c <= 0) return CS_EINVAL;
  if ((c & 3) || (ldx & 3) || (ldy & 3) || ldx < c || ldy < c) return CS_EINVAL;
  if (((uintptr_t)x & 15) || ((uintptr_t)y & 15) || ((uintptr_t)gamma & 15) ||
      ((uintptr_t)beta & 15))
    return CS_EINVAL;
  const int ch4 = c >> 2;
  const int grid = cs_grid_for(((int64_t)m + 3) / 4, 1, 256 * 32);
  hipStream_t s = (hipStream_t)stream;
  if (ch4 <= 64 * 2)
    CS_LAUNCH(ln_kernel<2>, dim3(grid), dim3(256), 0, s, x, gamma, beta, y, m, c, ldx, ldy, eps);
  else if (ch4 <= 64 * 4)
    CS_LAUNCH(ln_kernel<4>, dim3(grid), dim3(256), 0, s, x, gamma, beta, y, m, c, ldx, ldy, eps);
  else if (ch4 <= 64 * 8)
    CS_LAUNCH(ln_kernel<8>, dim3(grid), dim3(256), 0, s, x, gamma, beta, y, m, c, ldx, ldy, eps);
  else
    return CS_EINVAL;
  CS_CHECK_LAUNCH();
  return CS_OK;
}

// LayerNorm whose output is the interleaved F16X3 operand pair of y * a_scale (see ln_pair_kernel): y has the bytes of an
// fp32 [m][ldy] tensor (ldy in floats, c and ldy multiples of 16).  Feeds cs_conv_gemm with a_format = 2.
extern "C" int cs_layernorm_pair16(const float* x, const float* gamma, const float* beta, void* y, int m, int c, int ldx,
                                   int ldy, float eps, float a_scale, int32_t* status, cs_stream_t stream) {
  if (!x || !gamma || !beta || !y || m <= 0 || c <= 0 || !(a_scale > 0.f)) return CS_EINVAL;
  if ((c & 15) || (ldx & 3) || (ldy & 15) || ldx < c || ldy < c) return CS_EINVAL;
  if (((uintptr_t)x & 15) || ((uintptr_t)y & 15) || ((uintptr_t)gamma & 15) || ((uintptr_t)beta & 15)) return CS_EINVAL;
  const int ch4 = c >> 2;
  const int grid = cs_grid_for(((int64_t)m + 3) / 4, 1, 256 * 32);
  hipStream_t s = (hipStream_t)stream;
  _Float16* yo = (_Float16*)y;
  if (ch4 <= 64 * 2)
    CS_LAUNCH(ln_pair_kernel<2>, dim3(grid), dim3(256), 0, s, x, gamma, beta, yo, m, c, ldx, ldy, eps, a_scale, status);
  else if (ch4 <= 64 * 4)
    CS_LAUNCH(ln_pair_kernel<4>, dim3(grid), dim3(256), 0, s, x, gamma, beta, yo, m, c, ldx, ldy, eps, a_scale, status);
  else if (ch4 <= 64 * 8)
    CS_LAUNCH(ln_pair_kernel<8>, dim3(grid), dim3(256), 0, s, x, gamma, beta, yo, m, c, ldx, ldy, eps, a_scale, status);
  else
    return CS_EINVAL;
  CS_CHECK_LAUNCH();
  return CS_OK;
}
