// pos_embed.hip -- first layer of the skip-propagation point encoder (gfx950).
//
// The reference feeds cat([points (P, d), box feature repeated over the points (P, F)]) * mask
// through fc_pos = Linear(d + F, N) (models/iscnet/modules/skip_propagation.py:55-66,
// layers.py:364-366).  The F box-feature channels are one vector per proposal and the 0/1
// mask is one scalar per point, so
//   out[r][n] = bias[n] + mask[r] * (sum_{j<d} x[r][j] * W[n][j] + group[r / rows_per_group][n])
// with group = box_feature . W[:, d:]^T computed once per proposal by the caller.  This kernel
// writes the (M, N) result in ONE pass (HBM-bound: 4*N B per row), straight into the strided
// buffer the encoder's first GEMM reads; the torch composition (K = d GEMM + addcmul_) moved
// three times as many bytes.
#include "common.h"

namespace {

constexpr int PE_THREADS = 256;
constexpr int PE_MAX_D = 8;
typedef float pe4 __attribute__((ext_vector_type(4)));

// Thread t of a workgroup owns columns 4t .. 4t+3 (+ 1024 per pass) for a slab of rows;
// its W rows and bias stay in registers.
__global__ __launch_bounds__(PE_THREADS) void pos_embed_kernel(
    int M, int N, int d, int ldx, int rows_per_group, int rows_per_block,
    const float *__restrict__ x, const float *__restrict__ mask, const float *__restrict__ W,
    int ldw, const float *__restrict__ bias, const float *__restrict__ group,
    float *__restrict__ out, int ldo, float next_scale, unsigned *status) {
  float omax = 0.f;
  const int r0 = blockIdx.x * rows_per_block;
  const int r1 = min(M, r0 + rows_per_block);
  for (int n = threadIdx.x * 4; n < N; n += PE_THREADS * 4) {
    float w[4][PE_MAX_D];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int j = 0; j < PE_MAX_D; ++j) w[c][j] = j < d ? W[(size_t)(n + c) * ldw + j] : 0.f;
    const pe4 b = *reinterpret_cast<const pe4 *>(bias + n);
    for (int r = r0; r < r1; ++r) {
      const float m = mask[r];
      const pe4 g = *reinterpret_cast<const pe4 *>(group + (size_t)(r / rows_per_group) * N + n);
      float xr[PE_MAX_D];
#pragma unroll
      for (int j = 0; j < PE_MAX_D; ++j) xr[j] = j < d ? x[(size_t)r * ldx + j] : 0.f;
      pe4 o;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < PE_MAX_D; ++j) acc = __builtin_fmaf(xr[j], w[c][j], acc);
        o[c] = __builtin_fmaf(m, acc + g[c], b[c]);
        omax = __builtin_fmaxf(omax, __builtin_fabsf(o[c]));
      }
      *reinterpret_cast<pe4 *>(out + (size_t)r * ldo + n) = o;
    }
  }
  // the consumer is a split-precision GEMM (csrc/gemm_f16x3.hip, activations scaled by 2^sa before
  // the f16 split): a value beyond 65504 / 2^sa would be silently saturated there -- say so
  if (omax * next_scale >= 65504.f) atomicOr(status, 4u);
}

}  // namespace

// x [M][ldx] (first d columns used, d <= 8), mask [M], W [N][ldw] (first d columns),
// bias [N], group [M / rows_per_group][N], out [M][ldo]; N % 4 == 0, 16-byte aligned out / bias / group.
RFD_API int rfd_pos_embed(int M, int N, int d, const float *x, int ldx, const float *mask,
                          const float *W, int ldw, const float *bias, const float *group,
                          int rows_per_group, float *out, int ldo, int sa, void *stream) {
  if (M <= 0 || N <= 0) return 0;
  if (d < 0 || d > PE_MAX_D || (N & 3) || (ldo & 3) || rows_per_group <= 0 ||
      ((uintptr_t)out & 15) || ((uintptr_t)bias & 15) || ((uintptr_t)group & 15)) {
    rfd_set_error("rfd_pos_embed: need d <= 8, N % 4 == 0, ldo % 4 == 0, 16-byte aligned out / bias / group",
                  hipErrorInvalidValue);
    return (int)hipErrorInvalidValue;
  }
  RfdWorkspace *ws;
  {
    int rc = rfd_get_workspace(&ws);
    if (rc) return rc;
  }
  const int rows_per_block = 64;
  hipLaunchKernelGGL(pos_embed_kernel, dim3(ceil_div(M, rows_per_block)), dim3(PE_THREADS), 0,
                     (hipStream_t)stream, M, N, d, ldx, rows_per_group, rows_per_block, x, mask, W,
                     ldw, bias, group, out, ldo, ldexpf(1.f, sa), rfd_status_word(ws, (hipStream_t)stream));
  RFD_CHECK_LAUNCH();
  return 0;
}
