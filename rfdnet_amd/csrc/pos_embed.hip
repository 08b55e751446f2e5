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


// ---- the same layer, written as frag rows (csrc/gemm_f16x3.hip "Fragment-ordered split activations") -----------
// out block (rb, kb) = relu(out[32 rb .. +31][32 kb .. +31]) 2^sa split into f16 (hi, lo), in the channel order of
// the consumer's matrix instruction.  A wave owns a 32-row block at a time and walks its N / 32 channel blocks; lane
// (row, half) produces the 16 channels of its row the block assigns to it.  W[:, :d] and the bias sit in LDS (all
// lanes of a half-wave read the same address: broadcast); HBM-bound on the 4 N bytes per row it writes.
typedef unsigned pe_u4 __attribute__((ext_vector_type(4)));
typedef _Float16 pe_h2 __attribute__((ext_vector_type(2)));

template <int D>
__global__ __launch_bounds__(PE_THREADS) void pos_embed_frag_kernel(
    int M, int N, int ldx, int rows_per_group, const float *__restrict__ x, const float *__restrict__ mask,
    const float *__restrict__ W, int ldw, const float *__restrict__ bias, const float *__restrict__ group,
    unsigned char *__restrict__ out, long rb_stride, float a_scale, unsigned *status) {
  extern __shared__ __attribute__((aligned(16))) float pe_lds[];          // [N][D] W, then [N] bias
  float *wl = pe_lds, *bl = pe_lds + (size_t)N * D;
  for (int i = threadIdx.x; i < N * D; i += PE_THREADS) wl[i] = W[(size_t)(i / D) * ldw + (i % D)];
  for (int i = threadIdx.x; i < N; i += PE_THREADS) bl[i] = bias[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int half = lane >> 5;
  const int nrb = M / 32, kbs = N / 32;
  unsigned amax16 = 0u;
  for (int rb = blockIdx.x * (PE_THREADS / 64) + wave; rb < nrb; rb += gridDim.x * (PE_THREADS / 64)) {
    const int r = 32 * rb + (lane & 31);
    float xr[D];
#pragma unroll
    for (int j = 0; j < D; ++j) xr[j] = x[(size_t)r * ldx + j];
    const float m = mask[r];
    const float *grow = group + (size_t)(r / rows_per_group) * N + 4 * half;
    unsigned char *dst = out + (size_t)rb * rb_stride + lane * 16;
    for (int kb = 0; kb < kbs; ++kb) {
      unsigned hw[8], lw[8];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c0 = 32 * kb + 8 * q + 4 * half;
        const pe4 g = *reinterpret_cast<const pe4 *>(grow + 32 * kb + 8 * q);
        const pe4 b = *reinterpret_cast<const pe4 *>(bl + c0);
        float o[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          float acc = 0.f;
#pragma unroll
          for (int j = 0; j < D; ++j) acc = __builtin_fmaf(xr[j], wl[(c0 + c) * D + j], acc);
          o[c] = __builtin_fmaf(m, acc + g[c], b[c]);          // the fp32 value rfd_pos_embed stores
          o[c] = (o[c] > 0.f ? o[c] : 0.f) * a_scale;
        }
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
          const pe_h2 h2 = __builtin_bit_cast(pe_h2, __builtin_amdgcn_cvt_pkrtz(o[e], o[e + 1]));
          const float r0 = o[e] - (float)h2[0], r1 = o[e + 1] - (float)h2[1];
          const unsigned h = __builtin_bit_cast(unsigned, h2);
          hw[2 * q + e / 2] = h;
          lw[2 * q + e / 2] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(r0, r1));
          unsigned mx;
          asm("v_pk_max_u16 %0, %1, %2" : "=v"(mx) : "v"(amax16), "v"(h));
          amax16 = mx;
        }
      }
      unsigned char *d = dst + (size_t)kb * 4096;
      *reinterpret_cast<pe_u4 *>(d) = pe_u4{hw[0], hw[1], hw[2], hw[3]};
      *reinterpret_cast<pe_u4 *>(d + 1024) = pe_u4{lw[0], lw[1], lw[2], lw[3]};
      *reinterpret_cast<pe_u4 *>(d + 2048) = pe_u4{hw[4], hw[5], hw[6], hw[7]};
      *reinterpret_cast<pe_u4 *>(d + 3072) = pe_u4{lw[4], lw[5], lw[6], lw[7]};
    }
  }
  // hi words are non-negative; cvt_pkrtz saturates at 65504 = 0x7bff
  if ((amax16 & 0xffffu) >= 0x7bffu || (amax16 >> 16) >= 0x7bffu) atomicOr(status, 4u);
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

// The same layer written as frag rows for rfd_gemm_f16x3_frag: out = first 4-KiB block of row block 0,
// rb_stride = bytes between 32-row blocks.  M % 32 == 0, N % 32 == 0, rows_per_group % 32 == 0, d <= 8.
RFD_API int rfd_pos_embed_frag(int M, int N, int d, const float *x, int ldx, const float *mask,
                               const float *W, int ldw, const float *bias, const float *group,
                               int rows_per_group, void *out, long rb_stride, int sa, void *stream) {
  if (M <= 0 || N <= 0) return 0;
  if (d < 1 || d > PE_MAX_D || (M % 32) || (N % 32) || rows_per_group <= 0 || (rows_per_group % 32) ||
      (rb_stride & 15) || ((uintptr_t)out & 15) || ((uintptr_t)group & 15)) {
    rfd_set_error("rfd_pos_embed_frag: need 1 <= d <= 8, M % 32, N % 32, rows_per_group % 32, 16-byte aligned out / group",
                  hipErrorInvalidValue);
    return (int)hipErrorInvalidValue;
  }
  RfdWorkspace *ws;
  {
    int rc = rfd_get_workspace(&ws);
    if (rc) return rc;
  }
  const int waves = PE_THREADS / 64;
  int blocks = ceil_div(M / 32, waves);
  const int cap = ws->num_cu * 8;                     // a few workgroups per CU; each stages W once
  if (blocks > cap) blocks = cap;
  unsigned *status = rfd_status_word(ws, (hipStream_t)stream);
  const float a_scale = ldexpf(1.f, sa);
  unsigned char *o = (unsigned char *)out;
#define PE_FRAG(D)                                                                                                \
  case D:                                                                                                         \
    hipLaunchKernelGGL(pos_embed_frag_kernel<D>, dim3(blocks), dim3(PE_THREADS), (size_t)N * (D + 1) * sizeof(float), \
                       (hipStream_t)stream, M, N, ldx, rows_per_group, x, mask, W, ldw, bias, group, o, rb_stride,  \
                       a_scale, status);                                                                          \
    break;
  switch (d) {
    PE_FRAG(1) PE_FRAG(2) PE_FRAG(3) PE_FRAG(4) PE_FRAG(5) PE_FRAG(6) PE_FRAG(7) PE_FRAG(8)
  }
#undef PE_FRAG
  RFD_CHECK_LAUNCH();
  return 0;
}
