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
  hipLaunchKernelGGL(group_concat_kernel,
                     dim3(ceil_div(mn, GP_THREADS), 1 + ceil_div(c, GP_CH), b),
                     dim3(GP_THREADS), 0, (hipStream_t)stream, c, n, m, nsample,
                     1.0f / radius, normalize, use_xyz, xyz, new_xyz, features, idx,
                     out, grouped_xyz_out);
  RFD_CHECK_LAUNCH();
  return 0;
}
