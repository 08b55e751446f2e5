// group_points.hip -- neighbourhood gather (+ fused QueryAndGroup epilogue).
//
// Replaces _ext-src/src/group_points_gpu.cu (K5 group_points_kernel :8-28,
// K6 group_points_grad_kernel :43-64).  The reference uses one block per scene
// with a thread per (channel, centre) and a serial nsample loop, i.e. strided
// reads AND writes.  Here lanes run along the contiguous (centre, sample) axis
// of the output, each thread keeps its neighbour index in a register and walks
// a chunk of channels: the index is read once, writes are coalesced, and the
// gathered reads hit a (C, N) table that is L2-resident at these sizes.
// HBM-bandwidth bound: 4*C*M*ns B written + 4*M*ns B of indices read.
//
// When the (C, N) table is small (N <= 2048: SA2..SA4 and vote aggregation) the
// gather is served from LDS instead of L2: a workgroup stages GL_CH feature rows
// (coalesced 16-B reads), then every thread turns four neighbour indices (one
// int4 load) into one 16-B store per channel.  The L2 -> L1 gather traffic
// (one 64-B sector per 4-B element, ~37 TB/s at B = 32) was the real limit of
// the global-gather version, not HBM; see DESIGN.md 3.3.
#include "common.h"

namespace {

constexpr int GP_THREADS = 256;
constexpr int GP_CH = 8;  // channels per thread

__global__ __launch_bounds__(GP_THREADS) void group_points_kernel(
    int c, int n, int mn /* npoints*nsample */, const float *__restrict__ points,
    const int *__restrict__ idx, float *__restrict__ out) {
  const int bi = blockIdx.z;
  const int jk = blockIdx.x * GP_THREADS + threadIdx.x;
  if (jk >= mn) return;
  const int ii = idx[(size_t)bi * mn + jk];
  const int l0 = blockIdx.y * GP_CH;
  const float *p = points + ((size_t)bi * c + l0) * n + ii;
  float *o = out + ((size_t)bi * c + l0) * mn + jk;
  const int lc = (c - l0) < GP_CH ? (c - l0) : GP_CH;
  float v[GP_CH];
#pragma unroll
  for (int l = 0; l < GP_CH; ++l) v[l] = l < lc ? p[(size_t)l * n] : 0.f;
#pragma unroll
  for (int l = 0; l < GP_CH; ++l)
    if (l < lc) o[(size_t)l * mn] = v[l];
}

__global__ __launch_bounds__(GP_THREADS) void group_points_grad_kernel(
    int c, int n, int mn, const float *__restrict__ grad_out,
    const int *__restrict__ idx, float *__restrict__ grad_points) {
  const int bi = blockIdx.z;
  const int jk = blockIdx.x * GP_THREADS + threadIdx.x;
  if (jk >= mn) return;
  const int ii = idx[(size_t)bi * mn + jk];
  const int l0 = blockIdx.y * GP_CH;
  const int lc = (c - l0) < GP_CH ? (c - l0) : GP_CH;
  for (int l = 0; l < lc; ++l)
    atomicAdd(grad_points + ((size_t)bi * c + l0 + l) * n + ii,
              grad_out[((size_t)bi * c + l0 + l) * mn + jk]);
}

// The three relative-position channels of one (centre, sample) element.
__device__ __forceinline__ void grouped_xyz(int bi, int jk, int ii, int n, int m, int ns, int ctot,
                                            float inv_radius, int normalize, int use_xyz,
                                            const float *__restrict__ xyz,
                                            const float *__restrict__ new_xyz,
                                            float *__restrict__ out,
                                            float *__restrict__ gxyz_out) {
  const int mn = m * ns;
  const int j = jk / ns;
  const float *q = xyz + ((size_t)bi * n + ii) * 3;
  const float *ctr = new_xyz + ((size_t)bi * m + j) * 3;
  float g[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    g[a] = q[a] - ctr[a];                    // grouped_xyz -= new_xyz (:335)
    // grouped_xyz /= radius (:337).  On a GPU tensor torch divides by a
    // Python scalar as x * (1.0f / radius) (ATen div kernel, CPU-scalar
    // fast path) -- that is what the reference executes, so do the same.
    if (normalize) g[a] = g[a] * inv_radius;
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    if (use_xyz) out[((size_t)bi * ctot + a) * mn + jk] = g[a];
    if (gxyz_out) gxyz_out[((size_t)bi * 3 + a) * mn + jk] = g[a];
  }
}

// LDS-staged gather: blockIdx.y = chunk of GL_CH channels, blockIdx.x = slab of the
// (centre, sample) axis.  `out` points at the first feature channel of scene 0 and
// consecutive scenes are ctot*mn floats apart (ctot = c for plain group_points).
constexpr int GL_CH = 8;
typedef float vf4 __attribute__((ext_vector_type(4)));
constexpr int GL_MAX_N = 2048;  // GL_CH * N * 4 B <= 64 KiB of LDS

// QueryAndGroup's position channels inside the LDS kernel: the channel-chunk workgroups of
// a slab take turns (iteration it -> chunk it % chunks), so the work is spread evenly and
// reuses the indices already in registers (xyz == nullptr: plain group_points).
struct XyzPart {
  const float *xyz, *new_xyz;
  float *out, *gxyz_out;  // `out` = channel 0 of scene 0 of the concatenated tensor
  int m, ns, ctot, normalize, use_xyz;
  float inv_radius;
};

__global__ __launch_bounds__(GP_THREADS) void group_lds_kernel(
    int c, int n, int mn, int slab, int ctot, const float *__restrict__ features,
    const int *__restrict__ idx, float *__restrict__ out, XyzPart xp) {
  extern __shared__ float tab[];
  const int bi = blockIdx.z;
  const int l0 = blockIdx.y * GL_CH;
  const int lc = (c - l0) < GL_CH ? (c - l0) : GL_CH;
  const float *src = features + ((size_t)bi * c + l0) * n;
  for (int i = threadIdx.x * 4; i < lc * n; i += GP_THREADS * 4)
    *reinterpret_cast<float4 *>(tab + i) = *reinterpret_cast<const float4 *>(src + i);
  __syncthreads();
  const int j0 = blockIdx.x * slab;
  const int j1 = (j0 + slab) < mn ? (j0 + slab) : mn;
  const int *ip = idx + (size_t)bi * mn;
  float *o = out + ((size_t)bi * ctot + l0) * mn;
  int turn = 0;
  for (int jk = j0 + threadIdx.x * 4; jk < j1; jk += GP_THREADS * 4) {
    const int4 ii = *reinterpret_cast<const int4 *>(ip + jk);
    if (xp.xyz) {
      if (turn == (int)blockIdx.y) {
        const int i4[4] = {ii.x, ii.y, ii.z, ii.w};
        vf4 g[3];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float *q = xp.xyz + ((size_t)bi * n + i4[e]) * 3;
          const float *ctr = xp.new_xyz + ((size_t)bi * xp.m + (jk + e) / xp.ns) * 3;
#pragma unroll
          for (int a = 0; a < 3; ++a) {
            float d = q[a] - ctr[a];
            if (xp.normalize) d = d * xp.inv_radius;  // see grouped_xyz
            g[a][e] = d;
          }
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          if (xp.use_xyz)
            *reinterpret_cast<vf4 *>(xp.out + ((size_t)bi * xp.ctot + a) * mn + jk) = g[a];
          if (xp.gxyz_out)
            *reinterpret_cast<vf4 *>(xp.gxyz_out + ((size_t)bi * 3 + a) * mn + jk) = g[a];
        }
      }
      if (++turn == (int)gridDim.y) turn = 0;
    }
    if (lc == GL_CH) {
#pragma unroll
      for (int l = 0; l < GL_CH; ++l) {
        const float *t = tab + l * n;
        vf4 v = {t[ii.x], t[ii.y], t[ii.z], t[ii.w]};
        __builtin_nontemporal_store(v, reinterpret_cast<vf4 *>(o + (size_t)l * mn + jk));
      }
    } else {
      for (int l = 0; l < lc; ++l) {
        const float *t = tab + l * n;
        vf4 v = {t[ii.x], t[ii.y], t[ii.z], t[ii.w]};
        __builtin_nontemporal_store(v, reinterpret_cast<vf4 *>(o + (size_t)l * mn + jk));
      }
    }
  }
}

inline bool lds_gather_ok(int c, int n, int mn) {
  return c > 0 && n <= GL_MAX_N && (n & 3) == 0 && (mn & 3) == 0;
}

// Slabs of the (centre, sample) axis so that the launch has >= ~1024 workgroups.
inline void launch_group_lds(int b, int c, int n, int mn, int ctot, const float *features,
                             const int *idx, float *out, const XyzPart &xp,
                             hipStream_t stream) {
  const int chunks = ceil_div(c, GL_CH);
  int slabs = ceil_div(1024, chunks * b);
  const int max_slabs = ceil_div(mn, GP_THREADS * 4);
  if (slabs > max_slabs) slabs = max_slabs;
  if (slabs < 1) slabs = 1;
  const int slab = ceil_div(ceil_div(mn, slabs), GP_THREADS * 4) * GP_THREADS * 4;
  hipLaunchKernelGGL(group_lds_kernel, dim3(ceil_div(mn, slab), chunks, b),
                     dim3(GP_THREADS), (size_t)GL_CH * n * sizeof(float), stream, c, n, mn, slab,
                     ctot, features, idx, out, xp);
}

// QueryAndGroup epilogue (pointnet2_utils.py:333-344) in one pass:
//   out[:, 0:3]  = (xyz[idx] - new_xyz) [/ radius]      (if use_xyz)
//   out[:, 3: ]  = features[:, idx]
// blockIdx.y == 0 handles the xyz channels, y >= 1 the feature chunks.
__global__ __launch_bounds__(GP_THREADS) void group_concat_kernel(
    int c, int n, int m, int ns, float inv_radius, int normalize, int use_xyz,
    const float *__restrict__ xyz, const float *__restrict__ new_xyz,
    const float *__restrict__ features, const int *__restrict__ idx,
    float *__restrict__ out, float *__restrict__ gxyz_out) {
  const int bi = blockIdx.z;
  const int mn = m * ns;
  const int jk = blockIdx.x * GP_THREADS + threadIdx.x;
  if (jk >= mn) return;
  const int ii = idx[(size_t)bi * mn + jk];
  const int co = use_xyz ? 3 : 0;  // channel offset of the features
  const int ctot = co + c;
  if (blockIdx.y == 0) {
    grouped_xyz(bi, jk, ii, n, m, ns, ctot, inv_radius, normalize, use_xyz, xyz, new_xyz, out,
                gxyz_out);
  } else {
    const int l0 = (blockIdx.y - 1) * GP_CH;
    const int lc = (c - l0) < GP_CH ? (c - l0) : GP_CH;
    const float *p = features + ((size_t)bi * c + l0) * n + ii;
    float *o = out + ((size_t)bi * ctot + co + l0) * mn + jk;
    float v[GP_CH];
#pragma unroll
    for (int l = 0; l < GP_CH; ++l) v[l] = l < lc ? p[(size_t)l * n] : 0.f;
#pragma unroll
    for (int l = 0; l < GP_CH; ++l)
      if (l < lc) o[(size_t)l * mn] = v[l];
  }
}

}  // namespace

RFD_API int group_points_kernel_wrapper(int b, int c, int n, int npoints,
                                        int nsample, const float *points,
                                        const int *idx, float *out,
                                        void *stream) {
  const int mn = npoints * nsample;
  if (b <= 0 || c <= 0 || mn <= 0) return 0;
  if (lds_gather_ok(c, n, mn))
    launch_group_lds(b, c, n, mn, c, points, idx, out, XyzPart{}, (hipStream_t)stream);
  else
    hipLaunchKernelGGL(group_points_kernel,
                       dim3(ceil_div(mn, GP_THREADS), ceil_div(c, GP_CH), b),
                       dim3(GP_THREADS), 0, (hipStream_t)stream, c, n, mn, points,
                       idx, out);
  RFD_CHECK_LAUNCH();
  return 0;
}

RFD_API int group_points_grad_kernel_wrapper(int b, int c, int n, int npoints,
                                             int nsample, const float *grad_out,
                                             const int *idx, float *grad_points,
                                             void *stream) {
  const int mn = npoints * nsample;
  if (b <= 0 || c <= 0 || mn <= 0) return 0;
  hipLaunchKernelGGL(group_points_grad_kernel,
                     dim3(ceil_div(mn, GP_THREADS), ceil_div(c, GP_CH), b),
                     dim3(GP_THREADS), 0, (hipStream_t)stream, c, n, mn,
                     grad_out, idx, grad_points);
  RFD_CHECK_LAUNCH();
  return 0;
}

RFD_API int rfd_group_concat(int b, int c, int n, int m, int nsample,
                             float radius, int normalize, int use_xyz,
                             const float *xyz, const float *new_xyz,
                             const float *features, const int *idx, float *out,
                             float *grouped_xyz_out, void *stream) {
  const int mn = m * nsample;
  if (b <= 0 || mn <= 0) return 0;
  if (!features) c = 0;
  const float inv_radius = 1.0f / radius;
  if (lds_gather_ok(c, n, mn)) {
    const int co = use_xyz ? 3 : 0;
    XyzPart xp{xyz, new_xyz, out, grouped_xyz_out, m, nsample, co + c, normalize, use_xyz,
               inv_radius};
    launch_group_lds(b, c, n, mn, co + c, features, idx, out + (size_t)co * mn, xp,
                     (hipStream_t)stream);
  } else {
    hipLaunchKernelGGL(group_concat_kernel,
                       dim3(ceil_div(mn, GP_THREADS), 1 + ceil_div(c, GP_CH), b),
                       dim3(GP_THREADS), 0, (hipStream_t)stream, c, n, m, nsample, inv_radius,
                       normalize, use_xyz, xyz, new_xyz, features, idx, out, grouped_xyz_out);
  }
  RFD_CHECK_LAUNCH();
  return 0;
}
