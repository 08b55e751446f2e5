// interpolate.hip -- three_nn + three_interpolate for gfx950.
//
// Replaces _ext-src/src/interpolate_gpu.cu (K7 three_nn_kernel :9-59,
// K8 three_interpolate_kernel :72-101, K9 three_interpolate_grad_kernel
// :116-143).  Problem sizes on this network are tiny (512x256, 1024x512), so
// the point is parallelism and launch count, not bandwidth: one wave-sized
// workgroup per 64 unknowns with the known set staged in LDS (the reference
// uses a single block per scene).
#include "common.h"

namespace {

constexpr int NN_THREADS = 64;
constexpr int NN_TILE = 1024;

__global__ __launch_bounds__(NN_THREADS) void three_nn_kernel(
    int n, int m, const float *__restrict__ unknown,
    const float *__restrict__ known, float *__restrict__ dist2,
    int *__restrict__ idx) {
  __shared__ float s_k[NN_TILE * 3];
  const int bi = blockIdx.y;
  unknown += (size_t)bi * n * 3;
  known += (size_t)bi * m * 3;
  dist2 += (size_t)bi * n * 3;
  idx += (size_t)bi * n * 3;
  const int j = blockIdx.x * NN_THREADS + threadIdx.x;
  const bool in = j < n;
  const float ux = in ? unknown[(size_t)j * 3 + 0] : 0.f;
  const float uy = in ? unknown[(size_t)j * 3 + 1] : 0.f;
  const float uz = in ? unknown[(size_t)j * 3 + 2] : 0.f;
  double best1 = 1e40, best2 = 1e40, best3 = 1e40;  // :27
  int besti1 = 0, besti2 = 0, besti3 = 0;
  for (int t0 = 0; t0 < m; t0 += NN_TILE) {
    const int tn = (m - t0) < NN_TILE ? (m - t0) : NN_TILE;
    __syncthreads();
    for (int f = threadIdx.x; f < tn * 3; f += NN_THREADS) s_k[f] = known[(size_t)t0 * 3 + f];
    __syncthreads();
    for (int p = 0; p < tn; ++p) {  // LDS broadcast reads
      const float x = s_k[p * 3 + 0], y = s_k[p * 3 + 1], z = s_k[p * 3 + 2];
      const float d = sumsq3(ux - x, uy - y, uz - z);  // :33
      const int k = t0 + p;
      if (d < best1) {  // :34-49 strict '<' cascade
        best3 = best2; besti3 = besti2;
        best2 = best1; besti2 = besti1;
        best1 = d;     besti1 = k;
      } else if (d < best2) {
        best3 = best2; besti3 = besti2;
        best2 = d;     besti2 = k;
      } else if (d < best3) {
        best3 = d;     besti3 = k;
      }
    }
  }
  if (in) {
    dist2[(size_t)j * 3 + 0] = (float)best1;
    dist2[(size_t)j * 3 + 1] = (float)best2;
    dist2[(size_t)j * 3 + 2] = (float)best3;
    idx[(size_t)j * 3 + 0] = besti1;
    idx[(size_t)j * 3 + 1] = besti2;
    idx[(size_t)j * 3 + 2] = besti3;
  }
}

// grid (ceil(n/256), c, b): lanes along n (coalesced write), weights/idx of a
// point are re-read per channel from L1/L2 (12+12 B, tiny).
__global__ __launch_bounds__(256) void three_interpolate_kernel(
    int c, int m, int n, const float *__restrict__ points,
    const int *__restrict__ idx, const float *__restrict__ weight,
    float *__restrict__ out) {
  const int bi = blockIdx.z, l = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const int *pi = idx + ((size_t)bi * n + j) * 3;
  const float *pw = weight + ((size_t)bi * n + j) * 3;
  const float *pp = points + ((size_t)bi * c + l) * m;
  // p1*w1 + p2*w2 + p3*w3 under the contraction contract (:98-99)
  float t = pp[pi[1]] * pw[1];
  t = __builtin_fmaf(pp[pi[0]], pw[0], t);
  t = __builtin_fmaf(pp[pi[2]], pw[2], t);
  out[((size_t)bi * c + l) * n + j] = t;
}

// Feature propagation up to the shared MLP in ONE kernel (PointnetFPModule.forward, pointnet2_modules.py:383-392):
//   dist = sqrt(dist2) (ThreeNN.forward, pointnet2_utils.py:124-125);  r_t = 1 / (dist_t + 1e-8);
//   w_t = r_t / ((r_0 + r_1) + r_2);   out[:c] = three_interpolate(points, idx, w);   out[c:] = skip
// every step in fp32 with correctly rounded sqrt / divisions, in the order torch evaluates the reference's
// expressions; the interpolation itself is three_interpolate_kernel's expression.  grid (ceil(n/256), c + cs, b).
__global__ __launch_bounds__(256) void three_interpolate_cat_kernel(
    int c, int cs, int m, int n, const float *__restrict__ points, const int *__restrict__ idx,
    const float *__restrict__ dist2, const float *__restrict__ skip, float *__restrict__ out) {
  const int bi = blockIdx.z, l = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  float *o = out + ((size_t)bi * (c + cs) + l) * n + j;
  if (l >= c) {                                       // the skip connection's channels: torch.cat([interpolated, skip], 1)
    *o = skip[((size_t)bi * cs + (l - c)) * n + j];
    return;
  }
  const int *pi = idx + ((size_t)bi * n + j) * 3;
  const float *pd = dist2 + ((size_t)bi * n + j) * 3;
  const float r0 = 1.0f / (__builtin_sqrtf(pd[0]) + 1e-8f), r1 = 1.0f / (__builtin_sqrtf(pd[1]) + 1e-8f),
              r2 = 1.0f / (__builtin_sqrtf(pd[2]) + 1e-8f);
  const float norm = (r0 + r1) + r2;
  const float w0 = r0 / norm, w1 = r1 / norm, w2 = r2 / norm;
  const float *pp = points + ((size_t)bi * c + l) * m;
  float t = pp[pi[1]] * w1;
  t = __builtin_fmaf(pp[pi[0]], w0, t);
  t = __builtin_fmaf(pp[pi[2]], w2, t);
  *o = t;
}

__global__ __launch_bounds__(256) void three_interpolate_grad_kernel(
    int c, int n, int m, const float *__restrict__ grad_out,
    const int *__restrict__ idx, const float *__restrict__ weight,
    float *__restrict__ grad_points) {
  const int bi = blockIdx.z, l = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const int *pi = idx + ((size_t)bi * n + j) * 3;
  const float *pw = weight + ((size_t)bi * n + j) * 3;
  float *gp = grad_points + ((size_t)bi * c + l) * m;
  const float g = grad_out[((size_t)bi * c + l) * n + j];
  atomicAdd(gp + pi[0], g * pw[0]);
  atomicAdd(gp + pi[1], g * pw[1]);
  atomicAdd(gp + pi[2], g * pw[2]);
}

}  // namespace

RFD_API int three_nn_kernel_wrapper(int b, int n, int m, const float *unknown,
                                    const float *known, float *dist2, int *idx,
                                    void *stream) {
  if (b <= 0 || n <= 0) return 0;
  hipLaunchKernelGGL(three_nn_kernel, dim3(ceil_div(n, NN_THREADS), b),
                     dim3(NN_THREADS), 0, (hipStream_t)stream, n, m, unknown,
                     known, dist2, idx);
  RFD_CHECK_LAUNCH();
  return 0;
}

RFD_API int three_interpolate_kernel_wrapper(int b, int c, int m, int n,
                                             const float *points,
                                             const int *idx,
                                             const float *weight, float *out,
                                             void *stream) {
  if (b <= 0 || c <= 0 || n <= 0) return 0;
  hipLaunchKernelGGL(three_interpolate_kernel, dim3(ceil_div(n, 256), c, b),
                     dim3(256), 0, (hipStream_t)stream, c, m, n, points, idx,
                     weight, out);
  RFD_CHECK_LAUNCH();
  return 0;
}

RFD_API int three_interpolate_grad_kernel_wrapper(int b, int c, int n, int m,
                                                  const float *grad_out,
                                                  const int *idx,
                                                  const float *weight,
                                                  float *grad_points,
                                                  void *stream) {
  if (b <= 0 || c <= 0 || n <= 0) return 0;
  hipLaunchKernelGGL(three_interpolate_grad_kernel,
                     dim3(ceil_div(n, 256), c, b), dim3(256), 0,
                     (hipStream_t)stream, c, n, m, grad_out, idx, weight,
                     grad_points);
  RFD_CHECK_LAUNCH();
  return 0;
}

// out [b][c + cs][n] = cat([three_interpolate(points [b][c][m], idx, weights(dist2)), skip [b][cs][n]], 1);
// dist2 / idx [b][n][3] as three_nn_kernel_wrapper returns them (SQUARED distances).  skip may be NULL with cs = 0.
RFD_API int rfd_three_interpolate_cat(int b, int c, int cs, int m, int n, const float *points, const int *idx,
                                      const float *dist2, const float *skip, float *out, void *stream) {
  if (b <= 0 || c + cs <= 0 || n <= 0) return 0;
  if (c < 0 || cs < 0 || (cs > 0 && !skip)) {
    rfd_set_error("rfd_three_interpolate_cat: channel counts / skip", hipErrorInvalidValue);
    return (int)hipErrorInvalidValue;
  }
  hipLaunchKernelGGL(three_interpolate_cat_kernel, dim3(ceil_div(n, 256), c + cs, b), dim3(256), 0,
                     (hipStream_t)stream, c, cs, m, n, points, idx, dist2, skip, out);
  RFD_CHECK_LAUNCH();
  return 0;
}
