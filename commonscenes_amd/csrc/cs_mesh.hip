// SDF -> triangle mesh by marching cubes on the MI355X: the step right after the sampler,
// model/diff_utils/util_3d.py:194-236 `sdf_to_mesh` (PyMCubes on the CPU, one object at a time, level 0.02 on 64^3).
//
// HBM-bound scan + compaction, three kernels per batch, no atomics, deterministic output order:
//   cs_mc_count   per voxel: which of its three +x/+y/+z grid edges cross the level (a vertex each) and how many
//                 triangles its cube emits (case table); per 4096-voxel block: the two totals;
//   cs_mc_vertices  in-block exclusive scan + block bases (the 64 block totals of the object) -> every crossing edge
//                 writes its vertex (linear interpolation in fp64 like PyMCubes, which converts the volume to double)
//                 and the voxel's packed {vertex offset, crossing flags};
//   cs_mc_faces   per cube: its triangles, vertex ids looked up through the owning voxels' packed offsets.
// Vertices are unique per grid edge (shared between the cubes around it), in voxel-raster order; faces in cube-raster
// order.  The caller owns every buffer and sizes the outputs from the block totals (one small read-back).
#include "cs_common.h"
#include "cs_mc_tables.h"

namespace {

constexpr int MC_BLOCK = 4096;     // voxels per block
constexpr int MC_THREADS = 256;
constexpr int MC_PER = MC_BLOCK / MC_THREADS;   // consecutive voxels per thread

__device__ __forceinline__ int mc_case(const float* __restrict__ s, int n, int i, int j, int k, float level) {
  int c = 0;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const float v = s[((int64_t)(i + cs_mc_corner[q][0]) * n + (j + cs_mc_corner[q][1])) * n + (k + cs_mc_corner[q][2])];
    c |= (v < level) << q;
  }
  return c;
}

// crossing flags of voxel (i,j,k)'s own +x/+y/+z edges (bits 0..2) and its cube's triangle count
__device__ __forceinline__ void mc_voxel(const float* __restrict__ s, int n, int v, float level, int table, int& flags,
                                         int& ntri) {
  const int k = v % n, j = (v / n) % n, i = v / (n * n);
  const bool in0 = s[v] < level;
  flags = 0;
  if (i + 1 < n) flags |= (in0 != (s[v + n * n] < level)) << 0;
  if (j + 1 < n) flags |= (in0 != (s[v + n] < level)) << 1;
  if (k + 1 < n) flags |= (in0 != (s[v + 1] < level)) << 2;
  ntri = (i + 1 < n && j + 1 < n && k + 1 < n) ? cs_mc_n_tris[table][mc_case(s, n, i, j, k, level)] : 0;
}

__global__ __launch_bounds__(MC_THREADS) void mc_count_kernel(const float* __restrict__ sdf, int n, int nvox,
                                                              int blocks_per_obj, float level, int table,
                                                              int32_t* __restrict__ block_sums) {
  const int obj = blockIdx.y, blk = blockIdx.x;
  const float* s = sdf + (int64_t)obj * nvox;
  int nv = 0, nt = 0;
  for (int q = 0; q < MC_PER; ++q) {
    const int v = blk * MC_BLOCK + threadIdx.x * MC_PER + q;
    if (v < nvox) {
      int f, t;
      mc_voxel(s, n, v, level, table, f, t);
      nv += __popc(f);
      nt += t;
    }
  }
  __shared__ int sv[MC_THREADS / 64], st[MC_THREADS / 64];
  for (int o = 32; o > 0; o >>= 1) {
    nv += __shfl_xor(nv, o, 64);
    nt += __shfl_xor(nt, o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    sv[threadIdx.x >> 6] = nv;
    st[threadIdx.x >> 6] = nt;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int a = 0, b = 0;
    for (int w = 0; w < MC_THREADS / 64; ++w) {
      a += sv[w];
      b += st[w];
    }
    block_sums[((int64_t)obj * blocks_per_obj + blk) * 2 + 0] = a;
    block_sums[((int64_t)obj * blocks_per_obj + blk) * 2 + 1] = b;
  }
}

// exclusive prefix of `mine` over the block's threads (thread order); returns it, total in *tot
__device__ __forceinline__ int block_exclusive(int mine, int* tot) {
  __shared__ int wsum[MC_THREADS / 64 + 1];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int inc = mine;
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  __syncthreads();
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  int base = 0;
  for (int q = 0; q < w; ++q) base += wsum[q];
  if (tot) {
    int a = 0;
    for (int q = 0; q < MC_THREADS / 64; ++q) a += wsum[q];
    *tot = a;
  }
  return base + inc - mine;
}

__global__ __launch_bounds__(MC_THREADS) void mc_vertices_kernel(const float* __restrict__ sdf, int n, int nvox,
                                                                 int blocks_per_obj, float level,
                                                                 const int32_t* __restrict__ block_sums,
                                                                 const int64_t* __restrict__ vert_base,
                                                                 float* __restrict__ verts, int32_t* __restrict__ voff,
                                                                 float vdiv, float shift) {
  const int obj = blockIdx.y, blk = blockIdx.x;
  const float* s = sdf + (int64_t)obj * nvox;
  int flags[MC_PER], cnt = 0;
  for (int q = 0; q < MC_PER; ++q) {
    const int v = blk * MC_BLOCK + threadIdx.x * MC_PER + q;
    int t = 0;
    flags[q] = 0;
    if (v < nvox) mc_voxel(s, n, v, level, 0, flags[q], t);      // (the flags do not depend on the table)
    cnt += __popc(flags[q]);
  }
  int off = block_exclusive(cnt, nullptr);
  for (int b = 0; b < blk; ++b) off += block_sums[((int64_t)obj * blocks_per_obj + b) * 2];   // <= 63 cached loads
  float* vo = verts + vert_base[obj] * 3;
  for (int q = 0; q < MC_PER; ++q) {
    const int v = blk * MC_BLOCK + threadIdx.x * MC_PER + q;
    if (v >= nvox) break;
    voff[(int64_t)obj * nvox + v] = off | (flags[q] << 24);
    if (!flags[q]) continue;
    const int k = v % n, j = (v / n) % n, i = v / (n * n);
    const double f0 = (double)s[v];
    const int stride[3] = {n * n, n, 1};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      if (flags[q] & (1 << a)) {
        const double f1 = (double)s[v + stride[a]];
        const double t = ((double)level - f0) / (f1 - f0);        // PyMCubes: linear interpolation along the edge
        double p[3] = {(double)i, (double)j, (double)k};
        p[a] += t;
        vo[(int64_t)off * 3 + 0] = (float)(p[0] / (double)vdiv + (double)shift);   // util_3d.py:218: verts / n_cell - .5
        vo[(int64_t)off * 3 + 1] = (float)(p[1] / (double)vdiv + (double)shift);   // in float64, then .float()
        vo[(int64_t)off * 3 + 2] = (float)(p[2] / (double)vdiv + (double)shift);
        ++off;
      }
    }
  }
}

__global__ __launch_bounds__(MC_THREADS) void mc_faces_kernel(const float* __restrict__ sdf, int n, int nvox,
                                                              int blocks_per_obj, float level, int table,
                                                              const int32_t* __restrict__ block_sums,
                                                              const int64_t* __restrict__ face_base,
                                                              const int32_t* __restrict__ voff,
                                                              int64_t* __restrict__ faces) {
  const int obj = blockIdx.y, blk = blockIdx.x;
  const float* s = sdf + (int64_t)obj * nvox;
  const int32_t* vf = voff + (int64_t)obj * nvox;
  int cases[MC_PER], cnt = 0;
  for (int q = 0; q < MC_PER; ++q) {
    const int v = blk * MC_BLOCK + threadIdx.x * MC_PER + q;
    cases[q] = 0;
    if (v < nvox) {
      const int k = v % n, j = (v / n) % n, i = v / (n * n);
      if (i + 1 < n && j + 1 < n && k + 1 < n) cases[q] = mc_case(s, n, i, j, k, level);
    }
    cnt += cs_mc_n_tris[table][cases[q]];
  }
  int off = block_exclusive(cnt, nullptr);
  for (int b = 0; b < blk; ++b) off += block_sums[((int64_t)obj * blocks_per_obj + b) * 2 + 1];
  int64_t* fo = faces + face_base[obj] * 3;
  for (int q = 0; q < MC_PER; ++q) {
    const int nt = cs_mc_n_tris[table][cases[q]];
    if (!nt) continue;
    const int v = blk * MC_BLOCK + threadIdx.x * MC_PER + q;
    for (int t = 0; t < nt; ++t) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int e = cs_mc_tri_table[table][cases[q]][3 * t + c];
        const int owner = v + (cs_mc_edge_owner[e][0] * n + cs_mc_edge_owner[e][1]) * n + cs_mc_edge_owner[e][2];
        const int axis = cs_mc_edge_owner[e][3];
        const int packed = vf[owner];
        const int fl = (packed >> 24) & 7;
        fo[(int64_t)(off + t) * 3 + c] = (int64_t)((packed & 0xFFFFFF) + __popc(fl & ((1 << axis) - 1)));
      }
    }
    off += nt;
  }
}

}  // namespace

extern "C" int cs_mc_blocks_per_object(int n) {
  if (n < 2 || n > 160) return 0;      // 3 n^3 vertex ids must fit the 24-bit field of the packed voxel word
  return (n * n * n + MC_BLOCK - 1) / MC_BLOCK;
}

extern "C" int cs_mc_count(const float* sdf, int nb, int n, float level, int table, int32_t* block_sums,
                           cs_stream_t stream) {
  if (!sdf || !block_sums || nb <= 0 || n < 2 || n > 160 || table < 0 || table >= CS_MC_NTABLES) return CS_EINVAL;
  const int nvox = n * n * n, bpo = cs_mc_blocks_per_object(n);
  CS_LAUNCH(mc_count_kernel, dim3(bpo, nb), dim3(MC_THREADS), 0, (hipStream_t)stream, sdf, n, nvox, bpo, level, table,
            block_sums);
  CS_CHECK_LAUNCH();
  return CS_OK;
}

extern "C" int cs_mc_emit(const float* sdf, int nb, int n, float level, int table, const int32_t* block_sums,
                          const int64_t* vert_base, const int64_t* face_base, float* verts, int64_t* faces,
                          int32_t* voxel_ws, float vert_div, float vert_shift, cs_stream_t stream) {
  if (!sdf || !block_sums || !vert_base || !face_base || !verts || !faces || !voxel_ws || nb <= 0 || n < 2 || n > 160 ||
      !(vert_div > 0.f) || table < 0 || table >= CS_MC_NTABLES)
    return CS_EINVAL;
  const int nvox = n * n * n, bpo = cs_mc_blocks_per_object(n);
  hipStream_t s = (hipStream_t)stream;
  CS_LAUNCH(mc_vertices_kernel, dim3(bpo, nb), dim3(MC_THREADS), 0, s, sdf, n, nvox, bpo, level, block_sums, vert_base,
            verts, voxel_ws, vert_div, vert_shift);
  CS_CHECK_LAUNCH();
  CS_LAUNCH(mc_faces_kernel, dim3(bpo, nb), dim3(MC_THREADS), 0, s, sdf, n, nvox, bpo, level, table, block_sums, face_base,
            voxel_ws, faces);
  CS_CHECK_LAUNCH();
  return CS_OK;
}
