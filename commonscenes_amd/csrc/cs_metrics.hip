// Point-cloud metric kernels of the evaluation scripts, MI355X-native:
//   * Chamfer backward            extension/chamfer.cu:155-185 (NmDistanceGradKernel x 2)
//   * approximate EMD             scripts/pytorch_structural_losses/src/approxmatch.cu:3-182 (approxmatchkernel),
//                                 :184-224 (matchcostkernel), :229-320 (matchcostgrad1/2kernel)
// The reference runs the auction of approxmatch as ONE 512-thread block per cloud pair (32 blocks in total) with the
// whole level loop inside; here every pass of every level is its own launch over (row tiles x batch) -- the three
// passes need a device-wide dependency anyway -- so a batch fills the 256 CUs, with the other cloud staged through
// LDS in 1024-point tiles (every lane reads the same LDS word per step: a broadcast).  Per-row sums run over the other
// cloud in index order, exactly the reference's order.  Chamfer's scatter (atomicAdd in the reference: order-dependent
// rounding) is a deterministic gather here.
#include "cs_common.h"

namespace {

constexpr int TILE = 1024;

__device__ __forceinline__ float sqd(float x1, float y1, float z1, float x2, float y2, float z2) {
  return (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1) + (z2 - z1) * (z2 - z1);
}

// ---------------------------------------------------------------------------------------------------------------------
// Chamfer backward.  grad_a[j] = 2 g1[j] (a_j - b_idx1[j])  +  sum_{k : idx2[k] == j} 2 g2[k] (a_j - b_k)
// (first term: a_j's own nearest neighbour, chamfer.cu:165-168; second: every b_k whose nearest neighbour is a_j,
// the -(g (x1 - x2)) scatter of the kernel's second launch, :169-171), k ascending.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void chamfer_grad_kernel(const float* __restrict__ a, const float* __restrict__ bpts,
                                                           const float* __restrict__ g_own, const int32_t* __restrict__ idx_own,
                                                           const float* __restrict__ g_oth, const int32_t* __restrict__ idx_oth,
                                                           float* __restrict__ grad_a, int n, int m) {
  __shared__ int sidx[TILE];
  __shared__ float sg[TILE];
  const int bi = blockIdx.y;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const float* A = a + (int64_t)bi * n * 3;
  const float* B = bpts + (int64_t)bi * m * 3;
  float x1 = 0, y1 = 0, z1 = 0, gx = 0, gy = 0, gz = 0;
  if (j < n) {
    x1 = A[3 * j], y1 = A[3 * j + 1], z1 = A[3 * j + 2];
    const int j2 = idx_own[(int64_t)bi * n + j];
    const float g = g_own[(int64_t)bi * n + j] * 2;
    gx = g * (x1 - B[3 * j2]);
    gy = g * (y1 - B[3 * j2 + 1]);
    gz = g * (z1 - B[3 * j2 + 2]);
  }
  for (int k0 = 0; k0 < m; k0 += TILE) {
    const int cnt = min(TILE, m - k0);
    __syncthreads();
    for (int e = threadIdx.x; e < cnt; e += blockDim.x) {
      sidx[e] = idx_oth[(int64_t)bi * m + k0 + e];
      sg[e] = g_oth[(int64_t)bi * m + k0 + e];
    }
    __syncthreads();
    if (j < n) {
      for (int k = 0; k < cnt; ++k) {
        if (sidx[k] == j) {                                   // rare: one gather from global per hit
          const float g = sg[k] * 2;
          const float* q = B + 3 * (k0 + k);
          gx += -(g * (q[0] - x1));
          gy += -(g * (q[1] - y1));
          gz += -(g * (q[2] - z1));
        }
      }
    }
  }
  if (j < n) {
    float* o = grad_a + ((int64_t)bi * n + j) * 3;
    o[0] = gx, o[1] = gy, o[2] = gz;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// approxmatch.  temp [b][(n+m)*2] = remainL[n] | remainR[m] | ratioL[n] | ratioR[m]; match [b][m][n].
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void emd_init_kernel(float* __restrict__ match, float* __restrict__ temp, int n, int m,
                                                       float multiL, float multiR) {
  const int bi = blockIdx.y;
  float* mt = match + (int64_t)bi * n * m;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < (int64_t)n * m; i += (int64_t)gridDim.x * blockDim.x)
    mt[i] = 0.f;
  float* t = temp + (int64_t)bi * (n + m) * 2;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n + m; i += gridDim.x * blockDim.x)
    t[i] = i < n ? multiL : multiR;
}

// pass 1 (approxmatch.cu:31-62): ratioL[k] = remainL[k] / (1e-9 + sum_l exp(level d(k,l)) remainR[l])
__global__ __launch_bounds__(256) void emd_ratio_l_kernel(const float* __restrict__ xyz1, const float* __restrict__ xyz2,
                                                          float* __restrict__ temp, int n, int m, float level) {
  __shared__ float buf[TILE * 4];
  const int bi = blockIdx.y, k = blockIdx.x * blockDim.x + threadIdx.x;
  float* t = temp + (int64_t)bi * (n + m) * 2;
  const float *remainL = t, *remainR = t + n;
  float* ratioL = t + n + m;
  const float* p1 = xyz1 + (int64_t)bi * n * 3;
  const float* p2 = xyz2 + (int64_t)bi * m * 3;
  float x1 = 0, y1 = 0, z1 = 0;
  if (k < n) x1 = p1[3 * k], y1 = p1[3 * k + 1], z1 = p1[3 * k + 2];
  float suml = 1e-9f;
  for (int l0 = 0; l0 < m; l0 += TILE) {
    const int lend = min(m, l0 + TILE) - l0;
    __syncthreads();
    for (int l = threadIdx.x; l < lend; l += blockDim.x) {
      buf[4 * l] = p2[3 * (l0 + l)], buf[4 * l + 1] = p2[3 * (l0 + l) + 1], buf[4 * l + 2] = p2[3 * (l0 + l) + 2];
      buf[4 * l + 3] = remainR[l0 + l];
    }
    __syncthreads();
    for (int l = 0; l < lend; ++l) {
      const float d = level * sqd(x1, y1, z1, buf[4 * l], buf[4 * l + 1], buf[4 * l + 2]);
      suml += __expf(d) * buf[4 * l + 3];
    }
  }
  if (k < n) ratioL[k] = remainL[k] / suml;
}

// pass 2 (:79-113): sumr = remainR[l] sum_k exp(level d) ratioL[k]; ratioR = min(remainR/(sumr+1e-9),1) remainR;
// remainR = max(0, remainR - sumr)
__global__ __launch_bounds__(256) void emd_ratio_r_kernel(const float* __restrict__ xyz1, const float* __restrict__ xyz2,
                                                          float* __restrict__ temp, int n, int m, float level) {
  __shared__ float buf[TILE * 4];
  const int bi = blockIdx.y, l = blockIdx.x * blockDim.x + threadIdx.x;
  float* t = temp + (int64_t)bi * (n + m) * 2;
  float* remainR = t + n;
  const float* ratioL = t + n + m;
  float* ratioR = t + n + m + n;
  const float* p1 = xyz1 + (int64_t)bi * n * 3;
  const float* p2 = xyz2 + (int64_t)bi * m * 3;
  float x2 = 0, y2 = 0, z2 = 0;
  if (l < m) x2 = p2[3 * l], y2 = p2[3 * l + 1], z2 = p2[3 * l + 2];
  float sumr = 0;
  for (int k0 = 0; k0 < n; k0 += TILE) {
    const int kend = min(n, k0 + TILE) - k0;
    __syncthreads();
    for (int k = threadIdx.x; k < kend; k += blockDim.x) {
      buf[4 * k] = p1[3 * (k0 + k)], buf[4 * k + 1] = p1[3 * (k0 + k) + 1], buf[4 * k + 2] = p1[3 * (k0 + k) + 2];
      buf[4 * k + 3] = ratioL[k0 + k];
    }
    __syncthreads();
    for (int k = 0; k < kend; ++k)
      sumr += __expf(level * sqd(buf[4 * k], buf[4 * k + 1], buf[4 * k + 2], x2, y2, z2)) * buf[4 * k + 3];
  }
  if (l < m) {
    const float r = remainR[l];
    sumr *= r;
    const float consumption = fminf(r / (sumr + 1e-9f), 1.0f);
    ratioR[l] = consumption * r;
    remainR[l] = fmaxf(0.0f, r - sumr);
  }
}

// pass 3 (:131-160): w = exp(level d) ratioL[k] ratioR[l]; match[l][k] += w; remainL[k] = max(0, remainL[k] - sum_l w)
__global__ __launch_bounds__(256) void emd_match_kernel(const float* __restrict__ xyz1, const float* __restrict__ xyz2,
                                                        float* __restrict__ match, float* __restrict__ temp, int n, int m,
                                                        float level) {
  __shared__ float buf[TILE * 4];
  const int bi = blockIdx.y, k = blockIdx.x * blockDim.x + threadIdx.x;
  float* t = temp + (int64_t)bi * (n + m) * 2;
  float* remainL = t;
  const float* ratioL = t + n + m;
  const float* ratioR = t + n + m + n;
  const float* p1 = xyz1 + (int64_t)bi * n * 3;
  const float* p2 = xyz2 + (int64_t)bi * m * 3;
  float* mt = match + (int64_t)bi * n * m;
  float x1 = 0, y1 = 0, z1 = 0, rl = 0;
  if (k < n) x1 = p1[3 * k], y1 = p1[3 * k + 1], z1 = p1[3 * k + 2], rl = ratioL[k];
  float suml = 0;
  for (int l0 = 0; l0 < m; l0 += TILE) {
    const int lend = min(m, l0 + TILE) - l0;
    __syncthreads();
    for (int l = threadIdx.x; l < lend; l += blockDim.x) {
      buf[4 * l] = p2[3 * (l0 + l)], buf[4 * l + 1] = p2[3 * (l0 + l) + 1], buf[4 * l + 2] = p2[3 * (l0 + l) + 2];
      buf[4 * l + 3] = ratioR[l0 + l];
    }
    __syncthreads();
    if (k < n) {
      for (int l = 0; l < lend; ++l) {
        const float w = __expf(level * sqd(x1, y1, z1, buf[4 * l], buf[4 * l + 1], buf[4 * l + 2])) * rl * buf[4 * l + 3];
        mt[(int64_t)(l0 + l) * n + k] += w;                   // lanes = consecutive k: coalesced
        suml += w;
      }
    }
  }
  if (k < n) remainL[k] = fmaxf(0.0f, remainL[k] - suml);
}

// matchcost (:184-224): out[i] = sum_{k<m} sum_{j<n} match[k][j] |xyz2_k - xyz1_j|.  One block per pair; thread t sums
// its j = t, t+256, ... over all k in order, then a fixed tree over the 256 partials (deterministic).
__global__ __launch_bounds__(256) void emd_cost_kernel(const float* __restrict__ xyz1, const float* __restrict__ xyz2,
                                                       const float* __restrict__ match, float* __restrict__ out, int n,
                                                       int m) {
  __shared__ float buf[TILE * 3];
  __shared__ float part[256];
  const int bi = blockIdx.x;
  const float* p1 = xyz1 + (int64_t)bi * n * 3;
  const float* p2 = xyz2 + (int64_t)bi * m * 3;
  const float* mt = match + (int64_t)bi * n * m;
  float sub = 0;
  for (int k0 = 0; k0 < m; k0 += TILE) {
    const int cnt = min(TILE, m - k0);
    __syncthreads();
    for (int e = threadIdx.x; e < cnt * 3; e += blockDim.x) buf[e] = p2[(int64_t)k0 * 3 + e];
    __syncthreads();
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
      const float x1 = p1[3 * j], y1 = p1[3 * j + 1], z1 = p1[3 * j + 2];
      for (int k = 0; k < cnt; ++k) {
        const float dx = buf[3 * k] - x1, dy = buf[3 * k + 1] - y1, dz = buf[3 * k + 2] - z1;
        sub += mt[(int64_t)(k0 + k) * n + j] * sqrtf(dx * dx + dy * dy + dz * dz);
      }
    }
  }
  part[threadIdx.x] = sub;
  for (int s = 128; s > 0; s >>= 1) {
    __syncthreads();
    if (threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
  }
  if (threadIdx.x == 0) out[bi] = part[0];
}

// matchcostgrad1 (:268-288): grad1[l] = sum_k match[k][l] (x1 - x2) / max(|x1 - x2|, 1e-10), k ascending
__global__ __launch_bounds__(256) void emd_grad1_kernel(const float* __restrict__ xyz1, const float* __restrict__ xyz2,
                                                        const float* __restrict__ match, float* __restrict__ grad1, int n,
                                                        int m) {
  __shared__ float buf[TILE * 3];
  const int bi = blockIdx.y, l = blockIdx.x * blockDim.x + threadIdx.x;
  const float* p1 = xyz1 + (int64_t)bi * n * 3;
  const float* p2 = xyz2 + (int64_t)bi * m * 3;
  const float* mt = match + (int64_t)bi * n * m;
  float x1 = 0, y1 = 0, z1 = 0, dx = 0, dy = 0, dz = 0;
  if (l < n) x1 = p1[3 * l], y1 = p1[3 * l + 1], z1 = p1[3 * l + 2];
  for (int k0 = 0; k0 < m; k0 += TILE) {
    const int cnt = min(TILE, m - k0);
    __syncthreads();
    for (int e = threadIdx.x; e < cnt * 3; e += blockDim.x) buf[e] = p2[(int64_t)k0 * 3 + e];
    __syncthreads();
    if (l < n) {
      for (int k = 0; k < cnt; ++k) {
        const float ex = x1 - buf[3 * k], ey = y1 - buf[3 * k + 1], ez = z1 - buf[3 * k + 2];
        const float d = mt[(int64_t)(k0 + k) * n + l] * rsqrtf(fmaxf(ex * ex + ey * ey + ez * ez, 1e-20f));
        dx += ex * d, dy += ey * d, dz += ez * d;
      }
    }
  }
  if (l < n) {
    float* o = grad1 + ((int64_t)bi * n + l) * 3;
    o[0] = dx, o[1] = dy, o[2] = dz;
  }
}

// matchcostgrad2 (:229-267): grad2[k] = sum_j match[k][j] (x2 - x1) / max(|x2 - x1|, 1e-10): one wave per k, lanes
// stride j (coalesced match row), fixed butterfly reduction
__global__ __launch_bounds__(256) void emd_grad2_kernel(const float* __restrict__ xyz1, const float* __restrict__ xyz2,
                                                        const float* __restrict__ match, float* __restrict__ grad2, int n,
                                                        int m) {
  const int bi = blockIdx.y;
  const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (k >= m) return;
  const float* p1 = xyz1 + (int64_t)bi * n * 3;
  const float* q = xyz2 + ((int64_t)bi * m + k) * 3;
  const float* row = match + (int64_t)bi * n * m + (int64_t)k * n;
  const float x2 = q[0], y2 = q[1], z2 = q[2];
  float sx = 0, sy = 0, sz = 0;
  for (int j = lane; j < n; j += 64) {
    const float ex = x2 - p1[3 * j], ey = y2 - p1[3 * j + 1], ez = z2 - p1[3 * j + 2];
    const float d = row[j] * rsqrtf(fmaxf(ex * ex + ey * ey + ez * ez, 1e-20f));
    sx += ex * d, sy += ey * d, sz += ez * d;
  }
  sx = wave_sum(sx), sy = wave_sum(sy), sz = wave_sum(sz);
  if (lane == 0) {
    float* o = grad2 + ((int64_t)bi * m + k) * 3;
    o[0] = sx, o[1] = sy, o[2] = sz;
  }
}

}  // namespace

extern "C" int cs_chamfer_backward(const float* xyz1, const float* xyz2, const float* grad_dist1, const float* grad_dist2,
                                   const int32_t* idx1, const int32_t* idx2, float* grad_xyz1, float* grad_xyz2, int b,
                                   int n, int m, cs_stream_t stream) {
  if (!xyz1 || !xyz2 || !grad_dist1 || !grad_dist2 || !idx1 || !idx2 || !grad_xyz1 || !grad_xyz2 || b <= 0 || n <= 0 ||
      m <= 0 || b > 65535)
    return CS_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  CS_LAUNCH(chamfer_grad_kernel, dim3((n + 255) / 256, b), dim3(256), 0, s, xyz1, xyz2, grad_dist1, idx1, grad_dist2, idx2,
            grad_xyz1, n, m);
  CS_CHECK_LAUNCH();
  CS_LAUNCH(chamfer_grad_kernel, dim3((m + 255) / 256, b), dim3(256), 0, s, xyz2, xyz1, grad_dist2, idx2, grad_dist1, idx1,
            grad_xyz2, m, n);
  CS_CHECK_LAUNCH();
  return CS_OK;
}

extern "C" int cs_emd_approxmatch(const float* xyz1, const float* xyz2, float* match, float* temp, int b, int n, int m,
                                  cs_stream_t stream) {
  if (!xyz1 || !xyz2 || !match || !temp || b <= 0 || n <= 0 || m <= 0 || b > 65535) return CS_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const float multiL = n >= m ? 1.f : (float)(m / n);           // integer ratios, approxmatch.cu:6-12
  const float multiR = n >= m ? (float)(n / m) : 1.f;
  CS_LAUNCH(emd_init_kernel, dim3(cs_grid_for((int64_t)n * m, 256, 1024), b), dim3(256), 0, s, match, temp, n, m, multiL,
            multiR);
  CS_CHECK_LAUNCH();
  for (int j = 7; j > -2; --j) {                                // nine levels, -4^7 ... -4^-1 (:24-29)
    const float level = -powf(4.0f, (float)j);
    CS_LAUNCH(emd_ratio_l_kernel, dim3((n + 255) / 256, b), dim3(256), 0, s, xyz1, xyz2, temp, n, m, level);
    CS_CHECK_LAUNCH();
    CS_LAUNCH(emd_ratio_r_kernel, dim3((m + 255) / 256, b), dim3(256), 0, s, xyz1, xyz2, temp, n, m, level);
    CS_CHECK_LAUNCH();
    CS_LAUNCH(emd_match_kernel, dim3((n + 255) / 256, b), dim3(256), 0, s, xyz1, xyz2, match, temp, n, m, level);
    CS_CHECK_LAUNCH();
  }
  return CS_OK;
}

extern "C" int cs_emd_matchcost(const float* xyz1, const float* xyz2, const float* match, float* out, int b, int n, int m,
                                cs_stream_t stream) {
  if (!xyz1 || !xyz2 || !match || !out || b <= 0 || n <= 0 || m <= 0) return CS_EINVAL;
  CS_LAUNCH(emd_cost_kernel, dim3(b), dim3(256), 0, (hipStream_t)stream, xyz1, xyz2, match, out, n, m);
  CS_CHECK_LAUNCH();
  return CS_OK;
}

extern "C" int cs_emd_matchcost_grad(const float* xyz1, const float* xyz2, const float* match, float* grad1, float* grad2,
                                     int b, int n, int m, cs_stream_t stream) {
  if (!xyz1 || !xyz2 || !match || !grad1 || !grad2 || b <= 0 || n <= 0 || m <= 0 || b > 65535) return CS_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  CS_LAUNCH(emd_grad1_kernel, dim3((n + 255) / 256, b), dim3(256), 0, s, xyz1, xyz2, match, grad1, n, m);
  CS_CHECK_LAUNCH();
  CS_LAUNCH(emd_grad2_kernel, dim3((m + 3) / 4, b), dim3(256), 0, s, xyz1, xyz2, match, grad2, n, m);
  CS_CHECK_LAUNCH();
  return CS_OK;
}
