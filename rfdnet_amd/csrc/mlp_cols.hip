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
// Exact fp32 (no split precision: these layers feed vote positions and box parameters).  A workgroup owns MLP_P
// points of one scene and ALL channels; activations stay in LDS between layers.  The kernel is bound by how many
// weight bytes a CU keeps in flight (a first version with one 4-byte load per lane and k ran at ~30 GB/s per CU:
// 42 us for the 512 -> 256 -> 256 MLP), so a lane owns FOUR consecutive output channels (one 16-byte load of the
// transposed, 4-padded weight row per k) and the k range is split over the thread groups that many channel quads
// leave free (256 / quads-rounded-up-to-a-power-of-two slices); the slices' partial sums meet in LDS, where bias,
// the fixed-order sum over slices and the ReLU finish a layer.
#include "common.h"

namespace {

constexpr int MLP_THREADS = 256;
constexpr int MLP_P = 8;              // points per workgroup
constexpr int MLP_CMAX = 1024;        // widest layer
constexpr int MLP_KU = 8;             // k rows fetched per batch and slice
typedef float mlp4 __attribute__((ext_vector_type(4)));

struct MlpArgs {
  int n_layers, N;
  int width[5];                        // C_0 .. C_L
  const float *wt[4];                  // [C_{i-1}][pad4(C_i)], zero padded
  const float *bias[4];                // [C_i]
  int relu[4];
  const float *x;
  float *y;
};

__global__ __launch_bounds__(MLP_THREADS) void mlp_cols_kernel(MlpArgs a) {
  __shared__ __attribute__((aligned(16))) float act[2][MLP_CMAX][MLP_P];          // 64 KiB
  __shared__ __attribute__((aligned(16))) float part[MLP_THREADS][4][MLP_P];      // 32 KiB: [slice x quad][channel][point]
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
    const int ldw = (cout + 3) & ~3;                  // padded row length of the transposed weights
    const bool last = l + 1 == a.n_layers;
    // output channels in blocks of up to 1024 / ... : quads of this pass, rounded up to a power of two (<= 256)
    for (int cb = 0; cb < cout; cb += 4 * MLP_THREADS) {        // (one pass for every width <= 1024)
      const int quads = (min(cout - cb, 4 * MLP_THREADS) + 3) >> 2;
      int qp = 1;
      while (qp < quads) qp <<= 1;
      const int slices = MLP_THREADS / qp;
      const int q = t & (qp - 1), slice = t / qp;
      float acc[4][MLP_P];
#pragma unroll
      for (int ch = 0; ch < 4; ++ch)
#pragma unroll
        for (int p = 0; p < MLP_P; ++p) acc[ch][p] = 0.f;
      if (q < quads) {
        const float *w = a.wt[l] + cb + 4 * q;
        for (int k0 = slice; k0 < cin; k0 += slices * MLP_KU) {
          mlp4 wv[MLP_KU], a0[MLP_KU], a1[MLP_KU];
#pragma unroll
          for (int j = 0; j < MLP_KU; ++j) {
            const int k = k0 + j * slices;
            const int kc = k < cin ? k : cin - 1;                 // past the end: any valid row, masked below
            wv[j] = *reinterpret_cast<const mlp4 *>(w + (size_t)kc * ldw) * (k < cin ? 1.f : 0.f);
            a0[j] = *reinterpret_cast<const mlp4 *>(&act[cur][kc][0]);
            a1[j] = *reinterpret_cast<const mlp4 *>(&act[cur][kc][4]);
          }
#pragma unroll
          for (int j = 0; j < MLP_KU; ++j)
#pragma unroll
            for (int ch = 0; ch < 4; ++ch)
#pragma unroll
              for (int p = 0; p < 4; ++p) {
                acc[ch][p] = __builtin_fmaf(wv[j][ch], a0[j][p], acc[ch][p]);
                acc[ch][4 + p] = __builtin_fmaf(wv[j][ch], a1[j][p], acc[ch][4 + p]);
              }
        }
      }
#pragma unroll
      for (int ch = 0; ch < 4; ++ch) {
        *reinterpret_cast<mlp4 *>(&part[t][ch][0]) = mlp4{acc[ch][0], acc[ch][1], acc[ch][2], acc[ch][3]};
        *reinterpret_cast<mlp4 *>(&part[t][ch][4]) = mlp4{acc[ch][4], acc[ch][5], acc[ch][6], acc[ch][7]};
      }
      __syncthreads();
      // finish: thread (channel, point half) adds bias + the slices' partial sums in slice order
      for (int i = t; i < 4 * quads * 2; i += MLP_THREADS) {
        const int cl = i >> 1, ph = i & 1;                         // channel within the pass, which 4 points
        const int c = cb + cl;
        if (c < cout) {
          const float bv = a.bias[l][c];
          mlp4 s = mlp4{bv, bv, bv, bv};
          for (int sl = 0; sl < slices; ++sl)
            s += *reinterpret_cast<const mlp4 *>(&part[sl * qp + (cl >> 2)][cl & 3][4 * ph]);
          if (a.relu[l]) {
#pragma unroll
            for (int e = 0; e < 4; ++e) s[e] = s[e] > 0.f ? s[e] : 0.f;
          }
          if (last) *reinterpret_cast<mlp4 *>(a.y + ((size_t)b * cout + c) * a.N + n0 + 4 * ph) = s;
          else *reinterpret_cast<mlp4 *>(&act[cur ^ 1][c][4 * ph]) = s;
        }
      }
      __syncthreads();
    }
    cur ^= 1;
  }
}

}  // namespace

// x [B][widths[0]][N] -> y [B][widths[n_layers]][N] through 1 .. 4 layers; wt[i] = W_i TRANSPOSED with its rows zero
// padded to a multiple of four ([C_{i-1}][(C_i + 3) & ~3], BatchNorm folded in, 16-byte aligned), bias[i] [C_i],
// relu[i] != 0: ReLU after layer i.  N % 8 == 0, every width <= 1024, x / y 16-byte aligned.
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
    if (i < n_layers && ((uintptr_t)wt[i] & 15)) {
      rfd_set_error("rfd_mlp_cols: transposed weights must be 16-byte aligned", hipErrorInvalidValue);
      return (int)hipErrorInvalidValue;
    }
  }
  hipLaunchKernelGGL(mlp_cols_kernel, dim3(N / MLP_P, B), dim3(MLP_THREADS), 0, (hipStream_t)stream, a);
  RFD_CHECK_LAUNCH();
  return 0;
}
