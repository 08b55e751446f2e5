// gemm_f16x3.hip -- fp32-class GEMM on the f16 matrix cores (split precision).
//
// C[M,N] = act(A)[M,K] . W[N,K]^T + bias[N] + gbias[m / rows_per_group, N] + R[M,N]
//          (every addend optional; optional ReLU on the A operand and on C)
//
// Used for the big per-point layers of the skip-propagation encoder
// (models/iscnet/modules/layers.py:340-392 ResnetPointnet / ResnetBlockFC, called
// from skip_propagation.py:49-82), which the reference runs as fp32 cuBLAS GEMMs.
// gfx950 has no TF32/xf32; its exact-f32 MFMA runs at the vector rate (157 TF).
// Here each product is three f16 MFMAs on (hi, lo) splits of both operands with
// fp32 accumulation (hi*hi + hi*lo + lo*hi): ~2^-20 relative error per product,
// i.e. fp32-class results at up to 1/3 of the 2.5 PF f16 rate.
//
// Tiling: 128 x 128 output tile per 256-thread workgroup, 4 waves as 2 x 2, each
// 64 x 64 = 2 x 2 v_mfma_f32_32x32x16_f16 tiles; BK = 32; two LDS stages.
//   * A (activations, fp32 in HBM) is read as float4, optionally rectified, split
//     into f16 hi/lo in registers and written to LDS directly in MFMA-fragment
//     order (lane-linear 16-byte slots => conflict-free ds_write/ds_read_b128).
//   * W is split and laid out in fragment order ONCE (rfd_gemm_pack_w); a tile's
//     16 KiB per k-iteration is contiguous and streams in with LDS-DMA.
//   * accumulator columns (lane & 31) map to n, so every store instruction of a
//     wave writes 128 contiguous bytes of C.
// Operands are pre-scaled by 2^sa / 2^sw (exact) to keep the lo parts in the
// normal f16 range; the epilogue multiplies by 2^-(sa+sw).
#include "common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int A_STAGE_BYTES = (BM / 32) * (BK / 16) * 2 * 1024;  // 4 rowblocks x 2 ksteps x (hi,lo) x 1 KiB = 16 KiB
constexpr int W_STAGE_BYTES = (BN / 32) * (BK / 16) * 2 * 1024;  // 16 KiB
constexpr int STAGE_BYTES = A_STAGE_BYTES + W_STAGE_BYTES;

// W[N][K] fp32 -> fragment stream [N/128][K/32][nb 4][ks 2][split 2][lane 64][8] f16
__global__ void gemm_pack_w_kernel(int N, int K, int sw, const float *__restrict__ W,
                                   _Float16 *__restrict__ packed) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)N * K * 2;
  if (e >= total) return;
  const int j = e & 7;
  const int lane = (e >> 3) & 63;
  const int split = (e >> 9) & 1;
  const int ks = (e >> 10) & 1;
  const int nb = (e >> 11) & 3;
  const size_t rest = e >> 13;
  const int kiter = (int)(rest % (K / BK));
  const int ntile = (int)(rest / (K / BK));
  const int n = ntile * BN + nb * 32 + (lane & 31);
  const int k = kiter * BK + ks * 16 + 8 * (lane >> 5) + j;
  const float w = ldexpf(W[(size_t)n * K + k], sw);
  const _Float16 hi = (_Float16)w;
  const _Float16 lo = (_Float16)(w - (float)hi);
  packed[e] = split == 0 ? hi : lo;
}

__device__ __forceinline__ f32x16 mfma(half8 a, half8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

struct Args {
  const float *A; int lda;
  const half8 *Wp;
  float *C; int ldc;
  const float *bias;            // [N] or null
  const float *gbias; int rows_per_group;  // [M/rpg][N] or null
  const float *R; int ldr;      // [M][ldr] or null
  int M, N, K;
  int relu_in, relu_out;
  float a_scale, out_scale;     // 2^sa, 2^-(sa+sw)
};

__global__ __launch_bounds__(256, 2) void gemm_f16x3_kernel(Args g) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE_BYTES];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;        // wave position in the 2 x 2 grid
  // n tiles fastest: the workgroups sharing an A tile run back to back (L2 reuse)
  const int ntiles = g.N / BN;
  const int ntile = blockIdx.x % ntiles, mtile = blockIdx.x / ntiles;
  const int m0 = mtile * BM, n0 = ntile * BN;
  const int kiters = g.K / BK;

  // A loader geometry: thread -> (row, 16-wide k half)
  const int arow = t >> 1, akh = t & 1;
  const float *aptr = g.A + (size_t)(m0 + arow) * g.lda + akh * 16;
  const half8 *wsrc = g.Wp + ((size_t)ntile * kiters) * 1024 + (size_t)wave * 4 * 64 + lane;  // 16 frags per (tile,kiter), 4 per wave

  auto load_a = [&](int kit, f32x4 (&r)[4]) {
    const f32x4 *p = reinterpret_cast<const f32x4 *>(aptr + (size_t)kit * BK);
#pragma unroll
    for (int q = 0; q < 4; ++q) r[q] = p[q];
  };
  auto store_a = [&](int stage, const f32x4 (&r)[4]) {
    // fragment (rowblock = arow>>5, kstep = akh): element (row, k') -> lane = (row&31) + 32*(k'>>3), j = k'&7
    unsigned char *base = smem + stage * STAGE_BYTES + (((arow >> 5) * 2 + akh) * 2) * 1024;
    unsigned hw[8], lw[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int e = 0; e < 4; e += 2) {
        float a0 = r[q][e] * g.a_scale, a1 = r[q][e + 1] * g.a_scale;
        if (g.relu_in) { a0 = a0 > 0.f ? a0 : 0.f; a1 = a1 > 0.f ? a1 : 0.f; }
        const half2v h2 = __builtin_bit_cast(half2v, __builtin_amdgcn_cvt_pkrtz(a0, a1));
        const float r0 = a0 - (float)h2[0], r1 = a1 - (float)h2[1];
        hw[q * 2 + e / 2] = __builtin_bit_cast(unsigned, h2);
        lw[q * 2 + e / 2] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(r0, r1));
      }
    }
    const int l0 = arow & 31;
    u32x4 *hi = reinterpret_cast<u32x4 *>(base);
    u32x4 *lo = reinterpret_cast<u32x4 *>(base + 1024);
    hi[l0] = u32x4{hw[0], hw[1], hw[2], hw[3]};
    hi[l0 + 32] = u32x4{hw[4], hw[5], hw[6], hw[7]};
    lo[l0] = u32x4{lw[0], lw[1], lw[2], lw[3]};
    lo[l0 + 32] = u32x4{lw[4], lw[5], lw[6], lw[7]};
  };
  auto dma_w = [&](int stage, int kit) {
    unsigned char *dst = smem + stage * STAGE_BYTES + A_STAGE_BYTES + wave * 4 * 1024;
    const half8 *src = wsrc + (size_t)kit * 1024;
#pragma unroll
    for (int f = 0; f < 4; ++f)
      __builtin_amdgcn_global_load_lds((gbl_void *)(src + f * 64), (lds_void *)(dst + f * 1024), 16, 0, 0);
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x16{0.f};

  f32x4 ar[4];
  load_a(0, ar);
  dma_w(0, 0);
  store_a(0, ar);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int kit = 0; kit < kiters; ++kit) {
    const int st = kit & 1;
    const bool more = kit + 1 < kiters;
    if (more) {
      load_a(kit + 1, ar);     // global -> registers, consumed after the MFMAs
      dma_w(st ^ 1, kit + 1);  // global -> LDS (other stage)
    }
    const unsigned char *sa = smem + st * STAGE_BYTES;
    const unsigned char *sw = sa + A_STAGE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      half8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const half8 *fa = reinterpret_cast<const half8 *>(sa + ((((wm * 2 + i) * 2 + ks) * 2) * 1024)) + lane;
        ah[i] = fa[0];
        al[i] = fa[64];
        const half8 *fb = reinterpret_cast<const half8 *>(sw + ((((wn * 2 + i) * 2 + ks) * 2) * 1024)) + lane;
        bh[i] = fb[0];
        bl[i] = fb[64];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = mfma(ah[i], bh[j], acc[i][j]);
          acc[i][j] = mfma(ah[i], bl[j], acc[i][j]);
          acc[i][j] = mfma(al[i], bh[j], acc[i][j]);
        }
    }
    if (more) store_a(st ^ 1, ar);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- epilogue: scale back, add bias / group bias / residual, ReLU, store
  const int half = lane >> 5, nl = lane & 31;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + (wn * 2 + j) * 32 + nl;
      const float bn = g.bias ? g.bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + (wm * 2 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        float v = __builtin_fmaf(acc[i][j][r], g.out_scale, bn);
        if (g.gbias) v += g.gbias[(size_t)(m / g.rows_per_group) * g.N + n];
        if (g.R) v += g.R[(size_t)m * g.ldr + n];
        if (g.relu_out) v = v > 0.f ? v : 0.f;
        g.C[(size_t)m * g.ldc + n] = v;
      }
    }
}

}  // namespace

RFD_API size_t rfd_gemm_packed_bytes(int N, int K) { return (size_t)N * K * 2 * sizeof(_Float16); }

// W [N][K] fp32 (device) -> packed (device).  N % 128 == 0, K % 32 == 0.
RFD_API int rfd_gemm_pack_w(int N, int K, int sw, const float *W, void *packed, void *stream) {
  if (N % BN || K % BK) { rfd_set_error("rfd_gemm_pack_w: N % 128 or K % 32", hipErrorInvalidValue); return (int)hipErrorInvalidValue; }
  const size_t total = (size_t)N * K * 2;
  hipLaunchKernelGGL(gemm_pack_w_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, N, K, sw, W, (_Float16 *)packed);
  RFD_CHECK_LAUNCH();
  return 0;
}

// C = act(A) W^T (+bias)(+gbias[m / rows_per_group])(+R), optional ReLU in/out.
// M % 128 == 0, N % 128 == 0, K % 32 == 0; lda/ldc/ldr in floats, lda % 4 == 0,
// A 16-byte aligned.  sa / sw: power-of-two operand scalings (sw as packed).
RFD_API int rfd_gemm_f16x3(int M, int N, int K, const float *A, int lda, const void *packed_w,
                           float *C, int ldc, const float *bias, const float *gbias,
                           int rows_per_group, const float *R, int ldr, int relu_in, int relu_out,
                           int sa, int sw, void *stream) {
  if (M <= 0) return 0;
  if (M % BM || N % BN || K % BK || (lda & 3)) {
    rfd_set_error("rfd_gemm_f16x3: shape not a multiple of the 128x128x32 tile", hipErrorInvalidValue);
    return (int)hipErrorInvalidValue;
  }
  Args g;
  g.A = A; g.lda = lda; g.Wp = (const half8 *)packed_w; g.C = C; g.ldc = ldc; g.bias = bias;
  g.gbias = gbias; g.rows_per_group = rows_per_group > 0 ? rows_per_group : 1; g.R = R; g.ldr = ldr;
  g.M = M; g.N = N; g.K = K; g.relu_in = relu_in; g.relu_out = relu_out;
  g.a_scale = ldexpf(1.f, sa); g.out_scale = ldexpf(1.f, -(sa + sw));
  hipLaunchKernelGGL(gemm_f16x3_kernel, dim3((M / BM) * (N / BN)), dim3(256), 0, (hipStream_t)stream, g);
  RFD_CHECK_LAUNCH();
  return 0;
}
