// mlp_cols.hip -- the small shared MLPs of the detection stage as ONE kernel each (gfx950).
//
//   y [b][C_L][n] = layer_L( ... relu(layer_1(x [b][C_0][n])) ... ),   layer_i(a) = W_i a + b_i,  optional ReLU
//
// for the channel-major (B, C, N) tensors of the reference's 1x1 convolutions with eval-mode BatchNorm folded into
// (W_i, b_i) by the caller: PointnetFPModule's mlp (pointnet2_modules.py:395-403: 512 -> 256 -> 256 on 512 / 1024
// points), VotingModule (vote_module.py:34-61: 256 -> 256 -> 256 -> 259 on 1024 seeds), ProposalModule's head
// (proposal_module.py:85-124: 128 -> 128 -> 128 -> 69 on 256 proposals).  The reference (and rounds 1-5 here) runs
// conv, BatchNorm and ReLU of every layer as separate launches of ~4.5 us on ~0.1 GFLOP -- twenty launches for the
// three heads; the arithmetic is nothing (0.4 GFLOP per scene), the launches and the round trips were the cost.
//
// Exact fp32: one fma chain per output, bias first then k ascending (no split precision here: these layers feed
// vote positions and box parameters).  A workgroup owns MLP_P points of one scene and ALL channels: activations
// stay in LDS between layers, a thread computes channels t, t + 256, ... for the tile's points, weights are read
// TRANSPOSED ([C_in][C_out]: consecutive threads = consecutive addresses) from L2.
#include "common.h"

namespace {

constexpr int MLP_THREADS = 256;
constexpr int MLP_P = 8;              // points per workgroup
constexpr int MLP_CMAX = 1024;        // widest layer
typedef float mlp4 __attribute__((ext_vector_type(4)));

struct MlpArgs {
  int n_layers, N;
  int width[5];                        // C_0 .. C_L
  const float *wt[4];                  // [C_{i-1}][C_i]
  const float *bias[4];                // [C_i]
  int relu[4];
  const float *x;
  float *y;
};

__global__ __launch_bounds__(MLP_THREADS) void mlp_cols_kernel(MlpArgs a) {
  __shared__ __attribute__((aligned(16))) float act[2][MLP_CMAX][MLP_P];
  const int t = threadIdx.x;
  const int n0 = blockIdx.x * MLP_P, b = blockIdx.y;
  const int c0 = a.width[0];
  const float *xb = a.x + (size_t)b * c0 * a.N + n0;
  for (int c = t; c < c0; c += MLP_THREADS) {
    const mlp4 *src = reinterpret_cast<const mlp4 *>(xb + (size_t)c * a.N);
    *reinterpret_cast<mlp4 *>(&act[0][c][0]) = src[0];
    *reinterpret_cast<mlp4 *>(&act[0][c][4]) = src[1];
  }
  __syncthreads();
  int cur = 0;
  for (int l = 0; l < a.n_layers; ++l) {
    const int cin = a.width[l], cout = a.width[l + 1];
    const bool last = l + 1 == a.n_layers;
    const float *wt = a.wt[l];
    for (int c = t; c < cout; c += MLP_THREADS) {
      float acc[MLP_P];
      const float bv = a.bias[l][c];
#pragma unroll
      for (int p = 0; p < MLP_P; ++p) acc[p] = bv;
      const float *w = wt + c;
#pragma unroll 8
      for (int k = 0; k < cin; ++k) {
        const float wv = w[(size_t)k * cout];
        const mlp4 a0 = *reinterpret_cast<const mlp4 *>(&act[cur][k][0]);
        const mlp4 a1 = *reinterpret_cast<const mlp4 *>(&act[cur][k][4]);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          acc[p] = __builtin_fmaf(wv, a0[p], acc[p]);
          acc[4 + p] = __builtin_fmaf(wv, a1[p], acc[4 + p]);
        }
      }
      if (a.relu[l]) {
#pragma unroll
        for (int p = 0; p < MLP_P; ++p) acc[p] = acc[p] > 0.f ? acc[p] : 0.f;
      }
      if (last) {
        mlp4 *dst = reinterpret_cast<mlp4 *>(a.y + ((size_t)b * cout + c) * a.N + n0);
        dst[0] = mlp4{acc[0], acc[1], acc[2], acc[3]};
        dst[1] = mlp4{acc[4], acc[5], acc[6], acc[7]};
      } else {
        *reinterpret_cast<mlp4 *>(&act[cur ^ 1][c][0]) = mlp4{acc[0], acc[1], acc[2], acc[3]};
        *reinterpret_cast<mlp4 *>(&act[cur ^ 1][c][4]) = mlp4{acc[4], acc[5], acc[6], acc[7]};
      }
    }
    __syncthreads();
    cur ^= 1;
  }
}

}  // namespace

// x [B][widths[0]][N] -> y [B][widths[n_layers]][N] through 1 .. 4 layers; wt[i] = W_i TRANSPOSED ([C_{i-1}][C_i],
// BatchNorm folded in), bias[i] [C_i], relu[i] != 0: ReLU after layer i.  N % 8 == 0, every width <= 1024, x / y
// 16-byte aligned.
RFD_API int rfd_mlp_cols(int B, int N, int n_layers, const int *widths, const float *const *wt, const float *const *bias,
                         const int *relu, const float *x, float *y, void *stream) {
  if (B <= 0 || N <= 0) return 0;
  if (n_layers < 1 || n_layers > 4 || (N % MLP_P) || ((uintptr_t)x & 15) || ((uintptr_t)y & 15)) {
    rfd_set_error("rfd_mlp_cols: 1 .. 4 layers, N % 8 == 0, 16-byte aligned x / y", hipErrorInvalidValue);
    return (int)hipErrorInvalidValue;
  }
  MlpArgs a;
  a.n_layers = n_layers; a.N = N; a.x = x; a.y = y;
  for (int i = 0; i <= n_layers; ++i) {
    if (widths[i] < 1 || widths[i] > MLP_CMAX) {
      rfd_set_error("rfd_mlp_cols: layer width outside 1 .. 1024", hipErrorInvalidValue);
      return (int)hipErrorInvalidValue;
    }
    a.width[i] = widths[i];
  }
  for (int i = 0; i < 4; ++i) {
    a.wt[i] = i < n_layers ? wt[i] : nullptr;
    a.bias[i] = i < n_layers ? bias[i] : nullptr;
    a.relu[i] = i < n_layers ? relu[i] : 0;
  }
  hipLaunchKernelGGL(mlp_cols_kernel, dim3(N / MLP_P, B), dim3(MLP_THREADS), 0, (hipStream_t)stream, a);
  RFD_CHECK_LAUNCH();
  return 0;
}
