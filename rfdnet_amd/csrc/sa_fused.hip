// sa_fused.hip -- one set-abstraction layer after the ball query, fused:
//   group -> centre-subtract / normalise -> concat -> 3 x [1x1 conv + BN(eval) + ReLU]
//   -> max over the nsample neighbours, never materialising the (C, M, nsample) tensor.
//
// Replaces PointnetSAModuleVotes.forward after its grouper's ball query
// (external/pointnet2_ops_lib/pointnet2_ops/pointnet2_modules.py:219-255, QueryAndGroup
// pointnet2_utils.py:333-344, build_shared_mlp :9-19) at inference.
//
// A wave owns 32 rows = (centre, neighbour) pairs.  D[channel, row] = W[channel, k] x[k, row]
// with the exact-fp32 v_mfma_f32_32x32x2_f32 (no split precision: 10 GFLOP per scene):
// A = weights (fragment-ordered, staged per layer in LDS), B = activations held in registers.
// The accumulator layout of a layer is the B layout of the next one (lane (row, h) holds
// channels 8q + 4h + e of each 32-channel block; the host packs the next layer's k-order as
// that delivery order), so bias + ReLU are applied in place and the activations never leave
// the lane.  The last layer is reduced over the rows of a centre with lane shuffles.
// BatchNorm (eval) is folded into W and the bias on the host.
#include "common.h"
#include "../../include/rfd_pointnet2.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int SA_LDS_FLOATS = 36 * 1024;   // 144 KiB: the largest packed layer (128 x 264)

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// One layer: acc[b] = sum_j W[32b.., korder(j, kh)] * bin[j]; packed W in LDS as
// [block b][j4 = j / 4][lane][4 floats].
template <int KJ, int NB>   // KJ = k-steps (pairs of input channels, multiple of 4), NB = 32-channel output blocks
__device__ __forceinline__ void layer(const float *s_w, const float (&bin)[KJ], f32x16 (&acc)[NB], int lane) {
#pragma unroll
  for (int b = 0; b < NB; ++b) acc[b] = f32x16{0.f};
  const f32x4 *w = reinterpret_cast<const f32x4 *>(s_w) + lane;
#pragma unroll
  for (int j4 = 0; j4 < KJ / 4; ++j4) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const f32x4 w4 = w[(b * (KJ / 4) + j4) * 64];
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[b] = mfma32(w4[e], bin[4 * j4 + e], acc[b]);
    }
  }
}

// bias + ReLU in place; afterwards acc[b][r] is the next layer's B operand of k-step 16b + r
template <int NB>
__device__ __forceinline__ void bias_relu(f32x16 (&acc)[NB], const float *__restrict__ bias, int half) {
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 bv = *reinterpret_cast<const f32x4 *>(bias + 32 * b + 8 * q + 4 * half);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v = acc[b][4 * q + e] + bv[e];
        acc[b][4 * q + e] = v > 0.f ? v : 0.f;
      }
    }
}

__device__ __forceinline__ void stage(float *s_w, const float *__restrict__ g_w, int n_floats, int t) {
  const f32x4 *src = reinterpret_cast<const f32x4 *>(g_w);
  f32x4 *dst = reinterpret_cast<f32x4 *>(s_w);
  for (int i = t; i < n_floats / 4; i += 256) dst[i] = src[i];
}

template <int KJ1, int C1, int C2, int C3>
__global__ __launch_bounds__(256) void sa_fused_kernel(
    int n, int m, int ns, int c_feat, float inv_radius, const float *__restrict__ xyz,
    const float *__restrict__ new_xyz, const float *__restrict__ features, const int *__restrict__ idx,
    const float *__restrict__ w1, const float *__restrict__ b1, const float *__restrict__ w2,
    const float *__restrict__ b2, const float *__restrict__ w3, const float *__restrict__ b3,
    float *__restrict__ out) {
  __shared__ __attribute__((aligned(16))) float s_w[SA_LDS_FLOATS];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int half = lane >> 5, nrow = lane & 31;
  const int bi = blockIdx.y;
  const int rows = m * ns;
  const int row = (blockIdx.x * 4 + wave) * 32 + nrow;          // (centre, neighbour), may run past the end
  const bool live = row < rows;
  const int centre = live ? row / ns : 0;

  // ---- layer-1 input, channel c = 2j + kh of this lane's row: (xyz[idx] - centre) [/ r], features
  float bin1[KJ1];
  {
    const int pi = live ? idx[((size_t)bi * m + centre) * ns + (row - centre * ns)] : 0;
    const float *px = xyz + ((size_t)bi * n + pi) * 3;
    const float *pc = new_xyz + ((size_t)bi * m + centre) * 3;
#pragma unroll
    for (int j = 0; j < KJ1; ++j) {
      const int c = 2 * j + half;
      float v = 0.f;
      if (c < 3) v = (px[c] - pc[c]) * inv_radius;
      else if (c - 3 < c_feat) v = features[((size_t)bi * c_feat + (c - 3)) * n + pi];
      bin1[j] = v;
    }
  }
  stage(s_w, w1, C1 * KJ1 * 2, t);
  __syncthreads();
  f32x16 a1[C1 / 32];
  layer<KJ1, C1 / 32>(s_w, bin1, a1, lane);
  bias_relu<C1 / 32>(a1, b1, half);
  __syncthreads();

  stage(s_w, w2, C2 * C1, t);
  __syncthreads();
  float bin2[C1 / 2];
#pragma unroll
  for (int j = 0; j < C1 / 2; ++j) bin2[j] = a1[j >> 4][j & 15];
  f32x16 a2[C2 / 32];
  layer<C1 / 2, C2 / 32>(s_w, bin2, a2, lane);
  bias_relu<C2 / 32>(a2, b2, half);
  __syncthreads();

  stage(s_w, w3, C3 * C2, t);
  __syncthreads();
  float bin3[C2 / 2];
#pragma unroll
  for (int j = 0; j < C2 / 2; ++j) bin3[j] = a2[j >> 4][j & 15];
  f32x16 a3[C3 / 32];
  layer<C2 / 2, C3 / 32>(s_w, bin3, a3, lane);
  bias_relu<C3 / 32>(a3, b3, half);
  __syncthreads();

  // ---- max over the neighbours of a centre: lanes nrow..nrow+ns-1 of the same half (post-ReLU
  // values are >= 0, so dead rows contribute 0 safely only if they are excluded: mask them)
  float *s_part = s_w;                                  // [wave][half][C3 / 2 channel slots] for ns = 64
  const int span = ns < 32 ? ns : 32;
#pragma unroll
  for (int b = 0; b < C3 / 32; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = live ? a3[b][r] : 0.f;
      for (int d = 1; d < span; d <<= 1) v = fmaxf(v, __shfl_xor(v, d));
      a3[b][r] = v;
    }
  if (ns == 64) {       // a centre spans two waves (2w, 2w+1): combine through LDS
    if (nrow == 0) {
#pragma unroll
      for (int b = 0; b < C3 / 32; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) s_part[(wave * 2 + half) * (C3 / 2) + 16 * b + r] = a3[b][r];
    }
    __syncthreads();
    if (nrow == 0 && !(wave & 1) && live) {
#pragma unroll
      for (int b = 0; b < C3 / 32; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = fmaxf(a3[b][r], s_part[((wave + 1) * 2 + half) * (C3 / 2) + 16 * b + r]);
          const int ch = 32 * b + 8 * (r >> 2) + 4 * half + (r & 3);
          out[((size_t)bi * C3 + ch) * m + centre] = v;
        }
    }
  } else if (live && (nrow % ns) == 0) {
#pragma unroll
    for (int b = 0; b < C3 / 32; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ch = 32 * b + 8 * (r >> 2) + 4 * half + (r & 3);
        out[((size_t)bi * C3 + ch) * m + centre] = a3[b][r];
      }
  }
}

}  // namespace

// Packed layer layout (built by the host, rfdnet_amd/sa_fused.py): [C / 32][KJ / 4][64 lanes][4],
// element (b, j4, lane, e) = W'[32b + (lane & 31)][korder(4 j4 + e, lane >> 5)], W' = BN-folded
// weight, korder of layer 1 = 2j + kh (zero beyond the real input width), of layers 2 / 3 =
// 32 (j >> 4) + 8 ((j & 15) >> 2) + 4 kh + (j & 3).
RFD_API int rfd_sa_fused(int b, int n, int m, int nsample, int c_feat, float radius, int normalize_xyz,
                         const float *xyz, const float *new_xyz, const float *features, const int *idx,
                         int c1, int c2, int c3, const float *w1, const float *b1, const float *w2,
                         const float *b2, const float *w3, const float *b3, float *out, void *stream) {
  if (b <= 0 || m <= 0) return 0;
  const int cin = 3 + c_feat;
  const int kj1 = ((cin + 7) / 8) * 4;        // k-steps of layer 1: channel pairs, padded to 4
  if (!(nsample == 16 || nsample == 32 || nsample == 64)) {
    rfd_set_error("rfd_sa_fused: nsample must be 16, 32 or 64", hipErrorInvalidValue);
    return (int)hipErrorInvalidValue;
  }
  const float inv_r = normalize_xyz ? 1.0f / radius : 1.0f;
  const int rows = m * nsample;
  const dim3 grid((rows + 127) / 128, b);
  hipStream_t s = (hipStream_t)stream;
#define RFD_SA_CASE(KJ1, C1, C2, C3)                                                                    \
  if (kj1 == KJ1 && c1 == C1 && c2 == C2 && c3 == C3) {                                                 \
    hipLaunchKernelGGL((sa_fused_kernel<KJ1, C1, C2, C3>), grid, dim3(256), 0, s, n, m, nsample, c_feat, \
                       inv_r, xyz, new_xyz, features, idx, w1, b1, w2, b2, w3, b3, out);                 \
    RFD_CHECK_LAUNCH();                                                                                  \
    return 0;                                                                                            \
  }
  RFD_SA_CASE(4, 64, 64, 128)        // SA1: 3 + 1 inputs
  RFD_SA_CASE(68, 128, 128, 256)     // SA2: 3 + 128
  RFD_SA_CASE(132, 128, 128, 256)    // SA3, SA4: 3 + 256
  RFD_SA_CASE(132, 128, 128, 128)    // vote aggregation: 3 + 256
#undef RFD_SA_CASE
  rfd_set_error("rfd_sa_fused: layer widths not instantiated", hipErrorInvalidValue);
  return (int)hipErrorInvalidValue;
}
