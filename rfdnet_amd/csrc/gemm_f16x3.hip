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
#include <stdlib.h>

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

// running maximum of |hi| words (two f16 per register): an activation that reaches the f16 limit
// (|a| 2^sa >= 65504: cvt_pkrtz saturates, the product is silently wrong where the reference's
// fp32 GEMM is not) raises bit 2 of the device status word, as the decoder does (occ_decoder.hip)
__device__ __forceinline__ unsigned amax_u16(unsigned acc, unsigned hiw, bool nonneg) {
  unsigned r;
  const unsigned a = nonneg ? hiw : (hiw & 0x7fff7fffu);
  asm("v_pk_max_u16 %0, %1, %2" : "=v"(r) : "v"(acc), "v"(a));
  return r;
}
__device__ __forceinline__ void flag_overflow(unsigned amax16, unsigned *status) {
  if ((amax16 & 0xffffu) >= 0x7bffu || (amax16 >> 16) >= 0x7bffu) atomicOr(status, 4u);
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
  int gbias_stride;                        // N, or 0 when gbias is the zero vector (row-owner kernel)
  const float *R; int ldr;      // [M][ldr] or null
  int M, N, K;
  int relu_in, relu_out;
  float a_scale, out_scale;     // 2^sa, 2^-(sa+sw)
  float *pool;                  // [M/rows_per_group][N] running max(0, C) per group, or null (row-owner kernel)
  int pool_signed;              // pool holds the plain max (caller initialises it to -inf) instead of max(0, C)
  unsigned *status;             // device status word (bit 2: an activation left the f16 range)
  // fragment-ordered split activations (gemm_rowsf_kernel): first 4-KiB block of a 32-row block, stride between row blocks
  const unsigned char *Af; long af_stride;
  unsigned char *Cf; long cf_stride;      // Cf may be null (pool only)
  const float *zeros;                     // RFD_ZEROS_FLOATS zeros (the frag kernel's pool epilogue: bias already in the accumulators)
  // frag kernel: the two additive vectors folded into the accumulator start (zeros when absent); a stride is the
  // distance between groups (0 for a plain bias)
  const float *cb1, *cb2; int cb1_stride, cb2_stride;
};

__global__ __launch_bounds__(256, 2) void gemm_f16x3_kernel(Args g) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE_BYTES];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;        // wave position in the 2 x 2 grid
  // the workgroups sharing an A tile (its n tiles) go to ONE XCD (own L2; workgroups are
  // dealt round-robin to the 8 XCDs), at consecutive dispatch slots
  const int ntiles = g.N / BN;
  int ntile, mtile;
  if ((g.M / BM) % 8 == 0) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    ntile = j % ntiles;
    mtile = (j / ntiles) * 8 + xcd;
  } else {
    ntile = blockIdx.x % ntiles;
    mtile = blockIdx.x / ntiles;
  }
  const int m0 = mtile * BM, n0 = ntile * BN;
  const int kiters = g.K / BK;

  unsigned amax16 = 0u;
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
        amax16 = amax_u16(amax16, hw[q * 2 + e / 2], false);
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

  flag_overflow(amax16, g.status);
  // ---- epilogue: scale back, add bias / group bias / residual, ReLU, store
  const int half = lane >> 5, nl = lane & 31;
  float omax = 0.f;
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
        omax = __builtin_fmaxf(omax, __builtin_fabsf(v));
        g.C[(size_t)m * g.ldc + n] = v;
      }
    }
  if (omax * g.a_scale >= 65504.f) atomicOr(g.status, 4u);    // would overflow the next layer's split
}


// ======================================================================================
// Row-owner kernel for the big shapes (M % 256 == 0, N % 256 == 0, K % 128 == 0).
//
// The 128 x 128 kernel above re-reads 32 KiB of operands per 3 MFLOP issued and
// both operands pass through LDS; at the encoder's shapes it is bound by the
// L2 -> CU traffic, not by the matrix pipe.  Here a workgroup (8 waves) owns a 256 x 256
// output tile and each WAVE 32 rows x 256 columns = 8 accumulator blocks:
//   * activations never touch LDS: lane (n, h) loads 16 consecutive k of ITS rows
//     (one contiguous 64-byte chunk per row and 32-wide k piece), rectifies / scales /
//     splits them in registers and uses them directly as B fragments -- the k order
//     inside a piece is whatever that makes it (the packed W uses the same order);
//   * W streams through a 4-slot LDS ring of 32-KiB pieces (LDS-DMA, three pieces
//     ahead);
//   * the epilogue transposes each wave's block through the idle ring so that every row
//     is written in 512-byte runs (row-scattered 16-byte stores ran at ~8 B/clk/CU).
// Per 32-wide k piece a CU moves 64 KiB (32 KiB x fp32, 32 KiB W) for 12.6 MFLOP issued.
constexpr int RM = 256, RN = 256, RK = 32;
constexpr int R_PIECE_BYTES = 32 * 1024;

// layout 2: [N/256][K/32][kstep 2][blk 8][split 2][lane 64][8] f16,
// n = 256 ntile + 32 blk + (lane & 31), k = 32 piece + 16 (lane >> 5) + 8 kstep + j
__global__ void gemm_pack_rows_kernel(int N, int K, int sw, const float *__restrict__ W,
                                      _Float16 *__restrict__ packed) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)N * K * 2;
  if (e >= total) return;
  const int j = e & 7;
  const int lane = (e >> 3) & 63;
  const int split = (e >> 9) & 1;
  const int blk = (e >> 10) & 7;
  const int kstep = (e >> 13) & 1;
  const size_t rest = e >> 14;
  const int piece = (int)(rest % (K / RK));
  const int ntile = (int)(rest / (K / RK));
  const int n = ntile * RN + blk * 32 + (lane & 31);
  const int k = piece * RK + 16 * (lane >> 5) + 8 * kstep + j;
  const float w = ldexpf(W[(size_t)n * K + k], sw);
  const _Float16 hi = (_Float16)w;
  const _Float16 lo = (_Float16)(w - (float)hi);
  packed[e] = split == 0 ? hi : lo;
}

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Epilogue of the row-owner kernels: each wave transposes its 32 x 256 block through 16 KiB
// of (now idle) LDS, 128 columns at a time, and writes 512 contiguous bytes per row
// (row-scattered 16-byte stores from the accumulator layout run at ~8 B/clk/CU); lane l of a
// row handles columns 4l..4l+3, so bias + group bias (never null here: the host passes a zero
// vector for an absent one) are loaded once per pass.  16-byte chunks are XOR-swizzled by the
// row (conflict-free on both sides, no padding).  Optional fused max-pool over the group.
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_max(float v) {
  const int o = __builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROW_MASK, 0xf, false);
  const float w = __int_as_float(o);
  return w > v ? w : v;
}

// Pool-only epilogue (C == NULL, no residual).  In the accumulator layout a lane is a ROW of
// the product (point lane & 31) and its 16 registers per block are columns, so the group max is
// a cross-lane reduction: five DPP max steps per register (xor 1, xor 2, half-mirror, mirror,
// row_bcast15) leave the max over the wave's 32 rows in lanes 16-31 / 48-63.  When the
// workgroup's 256 rows lie in one group the eight waves then combine through LDS and thread c
// finishes column c (bias, scale, ReLU, ONE device-scope atomic per column and workgroup);
// otherwise every wave finishes its own columns.  The atomics cross the XCDs' private L2s and
// were the limit of these launches (8.4 M per K = 128 layer), not the arithmetic.
// max commutes with the monotone fma / ReLU, so the result is bit-identical to rows_epilogue's.
__device__ __forceinline__ void pool_finish(const Args &g, float v, size_t grp, int c) {
  float o = __builtin_fmaf(v, g.out_scale, g.bias[c] + g.gbias[grp * g.gbias_stride + c]);
  if (g.relu_out) o = o > 0.f ? o : 0.f;
  // the pooled value is the next split GEMM's input: same range watch as the stored outputs
  if (__builtin_fabsf(o) * g.a_scale >= 65504.f) atomicOr(g.status, 4u);
  int *p = reinterpret_cast<int *>(g.pool + grp * g.N + c);
  if (o > 0.f) atomicMax(p, __float_as_int(o));
  else if (g.pool_signed) {
    if (o == 0.f) atomicMax(p, 0);          // +-0 -> +0
    else atomicMin(reinterpret_cast<unsigned *>(p), __float_as_uint(o));
  }
}

__device__ __forceinline__ void pool_epilogue(const Args &g, unsigned char *smem, const f32x16 (&acc)[8], int wave,
                                              int lane, int m0, int n0) {
  const int j = lane & 15, half = lane >> 5;
  const int col = (j & 3) + 8 * (j >> 2) + 4 * half;
  const bool whole_wg = g.rows_per_group % RM == 0;     // uniform
  float *red = reinterpret_cast<float *>(smem);          // [8 waves][256 columns]
  const int qp = lane & 3, qi = (lane >> 2) & 3;
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    // max over the 32 rows (lanes of a half-wave) of each of the block's 16 registers, as a transposing reduction:
    // after the two quad steps a quad is uniform, so four registers share one (lane & 3 picks); after the two
    // rotations by 4 and 8 lanes a 16-lane row is uniform per quad position, so the four share one again
    // ((lane >> 2) & 3 picks): lane j of a row ends up with register j, exactly where the one-register-at-a-time
    // butterfly of rounds 1-5 left it (max is associative and commutative: bit-identical), in 57 instead of 96
    // instructions per block and without dependent chains (the DPP hazard no-ops are gone).
    float p[4];
#pragma unroll
    for (int h8 = 0; h8 < 2; ++h8) {                                    // eight registers at a time (register pressure)
      float v[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] = dpp_max<0xB1>(acc[b][8 * h8 + r]);      // quad_perm [1,0,3,2]
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] = dpp_max<0x4E>(v[r]);                    // quad_perm [2,3,0,1]
#pragma unroll
      for (int gq = 0; gq < 2; ++gq) {
        const float lo = qp & 1 ? v[4 * gq + 1] : v[4 * gq], hi = qp & 1 ? v[4 * gq + 3] : v[4 * gq + 2];
        p[2 * h8 + gq] = qp & 2 ? hi : lo;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) p[gq] = dpp_max<0x124>(p[gq]);      // row_ror:4
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) p[gq] = dpp_max<0x128>(p[gq]);      // row_ror:8
    const float lo = qi & 1 ? p[1] : p[0], hi = qi & 1 ? p[3] : p[2];
    float sel = qi & 2 ? hi : lo;                                       // register (lane & 15) of this block
    // the two 16-lane rows of a half-wave, lane by lane (the rows are no longer uniform, so row_bcast15 cannot be
    // used): ds_swizzle in bit mode, xor 16 within groups of 32 (no memory access: the LDS crossbar only)
    const float other = __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(sel), 0x401F));
    sel = other > sel ? other : sel;
    if (lane & 16) {
      if (whole_wg) red[wave * 256 + 32 * b + col] = sel;
      else pool_finish(g, sel, (size_t)(m0 / g.rows_per_group), n0 + 32 * b + col);
    }
  }
  if (whole_wg) {
    __syncthreads();
    const int t = threadIdx.x;
    if (t < 256) {
      float v = red[t];
#pragma unroll
      for (int w = 1; w < 8; ++w) v = red[w * 256 + t] > v ? red[w * 256 + t] : v;
      pool_finish(g, v, (size_t)((m0 - wave * 32) / g.rows_per_group), n0 + t);
    }
  }
}

template <bool HAS_RES>
__device__ __forceinline__ void rows_epilogue(const Args &g, unsigned char *smem, const f32x16 (&acc)[8], int wave,
                                              int lane, int m0, int n0) {
  if (!HAS_RES && !g.C) {
    pool_epilogue(g, smem, acc, wave, lane, m0, n0);
    return;
  }
  const int half = lane >> 5, n = lane & 31;
  {
    float *tr = reinterpret_cast<float *>(smem + wave * 16384);
    const int l = lane & 31, rsel = lane >> 5;
    const float *grow = g.gbias + (size_t)(m0 / g.rows_per_group) * g.gbias_stride + n0 + 4 * l;
    const float *brow = g.bias + n0 + 4 * l;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const f32x4 cb = *reinterpret_cast<const f32x4 *>(brow + 128 * p) +
                       *reinterpret_cast<const f32x4 *>(grow + 128 * p);
#pragma unroll
      for (int bb = 0; bb < 4; ++bb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int chunk = 8 * bb + 2 * q + half;
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[4 * p + bb][4 * q + e];
          *reinterpret_cast<f32x4 *>(tr + n * 128 + ((chunk ^ n) * 4)) = v;
        }
      const float pinit = g.pool_signed ? -__builtin_inff() : 0.f;
      f32x4 pmax = {pinit, pinit, pinit, pinit};
      float omax = 0.f;
#pragma unroll 8
      for (int j = 0; j < 16; ++j) {
        const int row = 2 * j + rsel;
        const f32x4 a = *reinterpret_cast<const f32x4 *>(tr + row * 128 + ((l ^ row) * 4));
        const size_t m = (size_t)(m0 + row);
        f32x4 add = cb;
        if (HAS_RES) add += *reinterpret_cast<const f32x4 *>(g.R + m * g.ldr + n0 + 128 * p + 4 * l);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[e] = __builtin_fmaf(a[e], g.out_scale, add[e]);
          if (g.relu_out) o[e] = o[e] > 0.f ? o[e] : 0.f;
          pmax[e] = o[e] > pmax[e] ? o[e] : pmax[e];
          omax = __builtin_fmaxf(omax, __builtin_fabsf(o[e]));
        }
        if (g.C) *reinterpret_cast<f32x4 *>(g.C + m * g.ldc + n0 + 128 * p + 4 * l) = o;   // C may be omitted when only the pool is wanted
      }
      // A STORED value this large cannot be split by the next layer (|a| 2^sa >= 65504).  The watch sits
      // on the outputs because this kernel's k loop is pinned instruction by instruction: one more VALU
      // op in it let the scheduler lift a conversion above its vmcnt wait (tools/audit_vmcnt.py).
      // (pool-only launches included: the pooled maximum feeds the next split GEMM as well)
      if (omax * g.a_scale >= 65504.f) atomicOr(g.status, 4u);
      if (g.pool) {
        // fused max-pool over the group's rows (every consumer rectifies the pooled vector, so
        // max(0, .) is what is needed): non-negative floats order like their bit patterns
        int *pp = reinterpret_cast<int *>(g.pool + (size_t)(m0 / g.rows_per_group) * g.N + n0 + 128 * p + 4 * l);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float other = __shfl_xor(pmax[e], 32);
          const float v = other > pmax[e] ? other : pmax[e];
          // non-negative floats order like their bit patterns as signed ints, negative ones in
          // reverse as unsigned ints; each atomic is a no-op against a stored value of the other sign
          if (rsel == 0) {
            if (v > 0.f) atomicMax(pp + e, __float_as_int(v));
            else if (g.pool_signed) {
              if (v == 0.f) atomicMax(pp + e, 0);          // +-0 -> +0
              else atomicMin(reinterpret_cast<unsigned *>(pp + e), __float_as_uint(v));
            }
          }
        }
      }
    }
  }
}

// 8 waves per workgroup: a wave owns 32 rows x 256 columns (8 accumulator blocks, <= 256
// registers), so TWO waves share each SIMD: while one is stuck issuing a vector-memory
// instruction, parked at a wait or in its epilogue, the other keeps the matrix pipe busy
// (vs 4 waves x 64 rows: +3 % on the loop-bound shapes, +23 % with a residual to read).
// The x loads are opaque asm (the compiler's own wait insertion would drain vmcnt and the
// LDS-DMA pipeline with it) and are only issued when their result is consumed: the
// compiler treats an asm's outputs as written at once and would otherwise reuse dead
// registers under data still in flight.
template <bool RELU_IN, bool HAS_RES>
__global__ __launch_bounds__(512) void gemm_rows8_kernel(Args g) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * R_PIECE_BYTES];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const unsigned lane16 = (unsigned)lane * 16u;
  const int half = lane >> 5, n = lane & 31;
  const int ntiles = g.N / RN;
  int ntile, mtile;
  {
    const int mtiles = g.M / RM;
    if (mtiles % 8 == 0) {
      const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
      ntile = j % ntiles;
      mtile = (j / ntiles) * 8 + xcd;
    } else {
      ntile = blockIdx.x % ntiles;
      mtile = blockIdx.x / ntiles;
    }
  }
  const int m0 = mtile * RM + wave * 32, n0 = ntile * RN;
  const int np = g.K / RK;
  const char *wp = reinterpret_cast<const char *>(g.Wp) +
                   ((size_t)g.N * g.K * 2 + (size_t)ntile * np * (R_PIECE_BYTES / 2)) * 2;
  const float *xp = g.A + (size_t)(m0 + n) * g.lda + 16 * half;

  f32x16 acc[8];
#pragma unroll
  for (int b = 0; b < 8; ++b) acc[b] = f32x16{0.f};
  f32x4 xr[2][4];
  half8 bh, bl;
  unsigned nh[4], nl[4];

  auto load_x2 = [&](int slot, int i) {   // loads 2i, 2i+1 of the 4 of a piece
    if (i == 0)
      asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:16"
                   : "=&v"(xr[slot][0]), "=&v"(xr[slot][1]) : "v"(xp) : "memory");
    else
      asm volatile("global_load_dwordx4 %0, %2, off offset:32\n\tglobal_load_dwordx4 %1, %2, off offset:48"
                   : "=&v"(xr[slot][2]), "=&v"(xr[slot][3]) : "v"(xp) : "memory");
  };
  auto dma = [&](unsigned poff, int slot, int jj) {
    const char *src = wp + poff + (wave * 4 + jj) * 1024;
    __builtin_amdgcn_global_load_lds((gbl_void *)(src + lane16),
                                     (lds_void *)(smem + slot * R_PIECE_BYTES + (wave * 4 + jj) * 1024), 16, 0, 0);
  };
  auto conv_slice = [&](int slot, int s, int i) {
    const f32x4 v = xr[slot][2 * s + (i >> 1)];
    float a0 = v[2 * (i & 1)] * g.a_scale, a1 = v[2 * (i & 1) + 1] * g.a_scale;
    if (RELU_IN) {
      a0 = a0 > 0.f ? a0 : 0.f;
      a1 = a1 > 0.f ? a1 : 0.f;
    }
    const half2v h2 = __builtin_bit_cast(half2v, __builtin_amdgcn_cvt_pkrtz(a0, a1));
    const float r0 = a0 - (float)h2[0], r1 = a1 - (float)h2[1];
    nh[i] = __builtin_bit_cast(unsigned, h2);
    nl[i] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(r0, r1));
  };
  auto conv_finish = [&]() {
    bh = __builtin_bit_cast(half8, u32x4{nh[0], nh[1], nh[2], nh[3]});
    bl = __builtin_bit_cast(half8, u32x4{nl[0], nl[1], nl[2], nl[3]});
  };

  load_x2(0, 0);
  load_x2(0, 1);
  xp += RK;
  load_x2(1, 0);
  load_x2(1, 1);
  xp += RK;
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) dma(p * R_PIECE_BYTES, p, jj);
  wait_vm<0>();
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int i = 0; i < 4; ++i) conv_slice(0, 0, i);
  conv_finish();

  const unsigned wbytes = (unsigned)np * R_PIECE_BYTES;
  unsigned doff = 3 * R_PIECE_BYTES;
  for (int p4 = 0; p4 < np; p4 += 4) {
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      const int xs = ps & 1;
      const bool load_next = p4 + ps + 2 < np;
      const half8 *w = reinterpret_cast<const half8 *>(smem + ps * R_PIECE_BYTES) + lane;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        half8 c[2][2], nx[2][2];
#pragma unroll
        for (int o = 0; o < 2; ++o) {
          c[o][0] = w[(s * 16 + 2 * o) * 64];
          c[o][1] = w[(s * 16 + 2 * o + 1) * 64];
        }
        const half8 h0 = bh, l0 = bl;
#pragma unroll
        for (int bp = 0; bp < 4; ++bp) {
          if (bp < 3) {
#pragma unroll
            for (int o = 0; o < 2; ++o) {
              nx[o][0] = w[(s * 16 + 4 * bp + 4 + 2 * o) * 64];
              nx[o][1] = w[(s * 16 + 4 * bp + 5 + 2 * o) * 64];
            }
          }
          const int b0 = 2 * bp, b1 = 2 * bp + 1;
          acc[b0] = mfma(c[0][0], h0, acc[b0]);
          acc[b1] = mfma(c[1][0], h0, acc[b1]);
          acc[b0] = mfma(c[0][0], l0, acc[b0]);
          acc[b1] = mfma(c[1][0], l0, acc[b1]);
          acc[b0] = mfma(c[0][1], h0, acc[b0]);
          acc[b1] = mfma(c[1][1], h0, acc[b1]);
          conv_slice(s == 0 ? xs : (xs ^ 1), s ^ 1, bp);
          if (s == 1 && load_next && bp < 2) load_x2(xs, bp);
          if (!(bp & 1)) dma(doff, (ps + 3) & 3, 2 * s + (bp >> 1));
#pragma unroll
          for (int q = 0; q < 6; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (q < 4) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            if (q == 3 && !(bp & 1)) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
          if (bp < 3) {
#pragma unroll
            for (int o = 0; o < 2; ++o) {
              c[o][0] = nx[o][0];
              c[o][1] = nx[o][1];
            }
          }
        }
        conv_finish();
        if (s == 0) wait_vm<3>();
      }
      if (load_next) xp += RK;
      doff = doff + R_PIECE_BYTES < wbytes ? doff + R_PIECE_BYTES : 0;
      __builtin_amdgcn_s_barrier();
    }
  }
  wait_vm<0>();
  __builtin_amdgcn_s_barrier();    // every wave's W transfers have landed: the ring is free
  rows_epilogue<HAS_RES>(g, smem, acc, wave, lane, m0, n0);
}


// ======================================================================================
// Fragment-ordered split activations ("frag rows"), round 6.
//
// The row-owner kernel above reads fp32 rows: lane = row, 32 distinct cache lines per load
// instruction, and every consumer (two n tiles per GEMM, two GEMMs per block input) rectifies,
// scales and splits the same activations again on the VALU.  Every consumer of an encoder
// activation rectifies it (layers.py:27,38-46: the in-place ReLU), so the PRODUCER can store
// relu(x) 2^sa already split into f16 (hi, lo) -- the same 4 bytes per element -- and store it in
// the order the consumer's matrix instruction wants its B operand:
//
//   block (rb, kb) = rows 32 rb .. +31, channels 32 kb .. +31 = 4 KiB:
//     [kstep 2][split hi / lo][lane 64][8 f16]
//     row     = 32 rb + (lane & 31)
//     channel = 32 kb + (r & 3) + 8 (r >> 2) + 4 (lane >> 5),   r = 8 kstep + j
//
// i.e. the channel order inside a block is the ACCUMULATOR order of v_mfma_f32_32x32x16 (a lane
// holds 16 channels of one row), so a producer's epilogue converts its accumulators in registers
// and writes four fully coalesced 1-KiB runs per block -- no transpose through LDS -- and the
// consumer's k loop issues four coalesced 16-byte loads per lane and 32-wide k piece whose results
// ARE the B fragments: no LDS, no VALU, no partial cache lines.  W is packed in the matching k
// order (layout 3 of rfd_gemm_pack_w).  A buffer is [M / 32][row-block stride]; a column window
// is a block offset, so [hidden | input] concatenations stay free.
constexpr int FRAG_BLOCK_BYTES = 4096;

__device__ __host__ __forceinline__ int frag_channel(int kstep, int half, int j) {
  const int r = 8 * kstep + j;
  return (r & 3) + 8 * (r >> 2) + 4 * half;
}

// layout 3: [N/256][K/32][kstep 2][blk 8][split 2][lane 64][8] f16,
// n = 256 ntile + 32 blk + (lane & 31), k = 32 piece + frag_channel(kstep, lane >> 5, j)
__global__ void gemm_pack_frag_kernel(int N, int K, int sw, const float *__restrict__ W,
                                      _Float16 *__restrict__ packed) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)N * K * 2;
  if (e >= total) return;
  const int j = e & 7;
  const int lane = (e >> 3) & 63;
  const int split = (e >> 9) & 1;
  const int blk = (e >> 10) & 7;
  const int kstep = (e >> 13) & 1;
  const size_t rest = e >> 14;
  const int piece = (int)(rest % (K / RK));
  const int ntile = (int)(rest / (K / RK));
  const int n = ntile * RN + blk * 32 + (lane & 31);
  const int k = piece * RK + frag_channel(kstep, lane >> 5, j);
  const float w = ldexpf(W[(size_t)n * K + k], sw);
  const _Float16 hi = (_Float16)w;
  const _Float16 lo = (_Float16)(w - (float)hi);
  packed[e] = split == 0 ? hi : lo;
}

// (hi, lo) words of two scaled, rectified values
__device__ __forceinline__ void split2(float a0, float a1, unsigned &hw, unsigned &lw) {
  const half2v h2 = __builtin_bit_cast(half2v, __builtin_amdgcn_cvt_pkrtz(a0, a1));
  const float r0 = a0 - (float)h2[0], r1 = a1 - (float)h2[1];
  hw = __builtin_bit_cast(unsigned, h2);
  lw = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(r0, r1));
}

// fp32 rows -> frag rows (relu optional; values scaled by 2^sa).  One thread = one lane of one block.
__global__ __launch_bounds__(256) void rows_to_frag_kernel(int M, int C, const float *__restrict__ x, int ldx,
                                                          int relu, float a_scale, unsigned char *__restrict__ out,
                                                          long rb_stride, unsigned *status) {
  const size_t u = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);      // block index (rb major)
  const int lane = threadIdx.x & 63;
  const int kbs = C / 32;
  if (u >= (size_t)(M / 32) * kbs) return;
  const int rb = (int)(u / kbs), kb = (int)(u % kbs);
  const float *row = x + (size_t)(32 * rb + (lane & 31)) * ldx + 32 * kb + 4 * (lane >> 5);
  unsigned char *dst = out + (size_t)rb * rb_stride + (size_t)kb * FRAG_BLOCK_BYTES + lane * 16;
  unsigned amax16 = 0u;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    unsigned hw[4], lw[4];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const f32x4 v = *reinterpret_cast<const f32x4 *>(row + 8 * (2 * s + q));
#pragma unroll
      for (int e = 0; e < 4; e += 2) {
        float a0 = v[e] * a_scale, a1 = v[e + 1] * a_scale;
        if (relu) { a0 = a0 > 0.f ? a0 : 0.f; a1 = a1 > 0.f ? a1 : 0.f; }
        split2(a0, a1, hw[2 * q + e / 2], lw[2 * q + e / 2]);
        amax16 = amax_u16(amax16, hw[2 * q + e / 2], false);
      }
    }
    *reinterpret_cast<u32x4 *>(dst + s * 2048) = u32x4{hw[0], hw[1], hw[2], hw[3]};
    *reinterpret_cast<u32x4 *>(dst + s * 2048 + 1024) = u32x4{lw[0], lw[1], lw[2], lw[3]};
  }
  flag_overflow(amax16, status);
}

// frag rows -> fp32 rows: (hi + lo) 2^-sa (exact: 22 significant bits)
__global__ __launch_bounds__(256) void frag_to_rows_kernel(int M, int C, const unsigned char *__restrict__ in,
                                                          long rb_stride, float inv_scale, float *__restrict__ x,
                                                          int ldx) {
  const size_t u = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int kbs = C / 32;
  if (u >= (size_t)(M / 32) * kbs) return;
  const int rb = (int)(u / kbs), kb = (int)(u % kbs);
  float *row = x + (size_t)(32 * rb + (lane & 31)) * ldx + 32 * kb + 4 * (lane >> 5);
  const unsigned char *src = in + (size_t)rb * rb_stride + (size_t)kb * FRAG_BLOCK_BYTES + lane * 16;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const half8 hi = *reinterpret_cast<const half8 *>(src + s * 2048);
    const half8 lo = *reinterpret_cast<const half8 *>(src + s * 2048 + 1024);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = ((float)hi[4 * q + e] + (float)lo[4 * q + e]) * inv_scale;
      *reinterpret_cast<f32x4 *>(row + 8 * (2 * s + q)) = v;
    }
  }
}

// Epilogue of the frag kernel.  bias + group bias are already IN the accumulators (gemm_rowsf_kernel starts them at
// (bias + gbias) 2^(sa+sw)), so an element is relu(acc) 2^-sw -> (hi, lo) -> four 1-KiB runs per 32 x 32 block, straight
// from the registers: no loads, no LDS.
__device__ __forceinline__ void frag_epilogue(const Args &g, const f32x16 (&acc)[8], int rb, int lane, int n0) {
  unsigned char *dst = g.Cf + (size_t)rb * g.cf_stride + (size_t)(n0 / 32) * FRAG_BLOCK_BYTES + lane * 16;
  const float post = g.out_scale * g.a_scale;                 // 2^-sw
  unsigned amax16 = 0u;
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    unsigned hw[8], lw[8];
#pragma unroll
    for (int w = 0; w < 8; ++w) {
      const float a0 = acc[b][2 * w], a1 = acc[b][2 * w + 1];
      split2((a0 > 0.f ? a0 : 0.f) * post, (a1 > 0.f ? a1 : 0.f) * post, hw[w], lw[w]);
      amax16 = amax_u16(amax16, hw[w], true);
    }
    unsigned char *d = dst + b * FRAG_BLOCK_BYTES;
    *reinterpret_cast<u32x4 *>(d) = u32x4{hw[0], hw[1], hw[2], hw[3]};
    *reinterpret_cast<u32x4 *>(d + 1024) = u32x4{lw[0], lw[1], lw[2], lw[3]};
    *reinterpret_cast<u32x4 *>(d + 2048) = u32x4{hw[4], hw[5], hw[6], hw[7]};
    *reinterpret_cast<u32x4 *>(d + 3072) = u32x4{lw[4], lw[5], lw[6], lw[7]};
  }
  // cvt_pkrtz saturates at 65504 = 0x7bff: a stored hi word that large means the value left the f16 range
  flag_overflow(amax16, g.status);
}

// Row-owner GEMM on frag rows, PERSISTENT: one workgroup per CU walks its list of 256 x 256 tiles (same n tile, so the
// same W stream, for all of them) and the operand streams never stop at a tile boundary -- the W ring keeps rolling
// (piece np of a tile IS piece 0 of the next: the transfer offset wraps) and the activation loads switch to the next
// tile's row block two pieces before the end.  Between two tiles only the epilogue runs (accumulators -> split -> four
// 1-KiB runs per block; pool), while the next tile's first pieces are already landing.  Measured on the one-tile-per-
// launch version of this kernel (tools/ab/r06_sessions.md): per 58-us tile the prologue fetch (x 2 pieces + W 3 pieces,
// ~4-5 us exposed), the useless wrap-around transfers at the end (~3 us at the LDS-DMA rate) and the workgroup
// hand-over cost more than the epilogue's arithmetic.
//
// Same tiling as gemm_rows8_kernel (a wave = 32 rows x 256 columns, W through the 4-slot LDS ring), but the activation
// operand arrives as ready B fragments.  Vector-memory LOADS per wave and piece, in issue order:
//   k step 0: dma, dma, [end] 2 loads (k-step-0 fragments of piece p + 2)
//   k step 1: dma, dma, [end] 2 loads (k-step-1 fragments of piece p + 2)
// always issued (the last tile of a workgroup re-reads its own first pieces: the pattern is what the waits count on).
// vmcnt retires loads in order, so `s_waitcnt vmcnt(N)` proves a load complete iff at least N loads were issued after
// it (stores in between only make the wait longer).  The fragments a k step consumes were issued a piece and a half
// earlier with exactly 12 loads behind them -> wait_vm<12> at the end of every k step; the W transfers of piece p + 1
// (issued during piece p - 2) have 14 behind them at the end of k step 0 of piece p, before the piece barrier.
template <bool STORE>
__global__ __launch_bounds__(512) void gemm_rowsf_kernel(Args g) {
  // ring (4 x 32 KiB) | per wave: the next tile's accumulator start values (8 x 1 KiB) | pool_epilogue's cross-wave
  // scratch (8 KiB; the ring is live during the epilogue).  ONE object: with separate __shared__ arrays the compiler
  // could no longer tell the ring's LDS-DMA writes from the reads and put `s_waitcnt vmcnt(0)` in front of every
  // ds_read of the loop
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * R_PIECE_BYTES + 16384];
  float (*cbs)[256] = reinterpret_cast<float (*)[256]>(smem + 4 * R_PIECE_BYTES);
  float *red = reinterpret_cast<float *>(smem + 4 * R_PIECE_BYTES + 8192);
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const unsigned lane16 = (unsigned)lane * 16u;
  const int ntiles = g.N / RN, mtiles = g.M / RM;
  // tile list of this workgroup: m tile (k0 + i kstep) mul + add, i = 0 .. n_my - 1, fixed n tile.  With 8 | m tiles
  // and (8 n tiles) | workgroups, the n tiles of one m tile run at the same time on ONE XCD (workgroups are dealt
  // round-robin to the XCDs; each has its own L2: the activation tile comes from HBM once).
  int ntile, k0, kstep, mul, add, kend;
  {
    const int nwg = gridDim.x, w = blockIdx.x;
    if (mtiles % 8 == 0 && nwg % (8 * ntiles) == 0) {
      const int j = w >> 3;
      ntile = j % ntiles;
      k0 = j / ntiles;
      kstep = (nwg >> 3) / ntiles;
      mul = 8;
      add = w & 7;
      kend = mtiles >> 3;
    } else {                                  // the launcher makes the workgroup count a multiple of n tiles
      ntile = w % ntiles;
      k0 = w / ntiles;
      kstep = nwg / ntiles;
      mul = 1;
      add = 0;
      kend = mtiles;
    }
  }
  if (k0 >= kend) return;
  const int n0 = ntile * RN;
  const int np = g.K / RK;
  const char *wp = reinterpret_cast<const char *>(g.Wp) +
                   ((size_t)g.N * g.K * 4 + (size_t)ntile * np * (R_PIECE_BYTES / 2)) * 2;
  const int half = lane >> 5;
  const float pre = 1.f / g.out_scale;

  f32x16 acc[8];
  u32x4 F[2][4];
  f32x4 cbv[2];
  const unsigned char *xp;

  auto load_k0 = [&](int slot) {
    asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:1024"
                 : "=&v"(F[slot][0]), "=&v"(F[slot][1]) : "v"(xp) : "memory");
  };
  auto load_k1 = [&](int slot) {
    asm volatile("global_load_dwordx4 %0, %2, off offset:2048\n\tglobal_load_dwordx4 %1, %2, off offset:3072"
                 : "=&v"(F[slot][2]), "=&v"(F[slot][3]) : "v"(xp) : "memory");
  };
  auto dma = [&](unsigned poff, int slot, int jj) {
    const char *src = wp + poff + (wave * 4 + jj) * 1024;
    __builtin_amdgcn_global_load_lds((gbl_void *)(src + lane16),
                                     (lds_void *)(smem + slot * R_PIECE_BYTES + (wave * 4 + jj) * 1024), 16, 0, 0);
  };
  // the additive vectors of a tile's group, 4 channels per lane (256 per wave), as two opaque loads
  auto load_cb = [&](int rb) {
    const size_t grp = (size_t)((rb * 32) / g.rows_per_group);
    const float *v1 = g.cb1 + grp * g.cb1_stride + n0 + 4 * lane;
    const float *v2 = g.cb2 + grp * g.cb2_stride + n0 + 4 * lane;
    asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %3, off"
                 : "=&v"(cbv[0]), "=&v"(cbv[1]) : "v"(v1), "v"(v2) : "memory");
  };
  // ... summed, scaled to the accumulator's units ((bias + gbias) 2^(sa+sw)) and parked in this wave's LDS row
  auto park_cb = [&]() {
    // the loads are opaque to the compiler: without this (volatile asm statements keep their order, so it stays behind
    // the s_waitcnt) the add below may be scheduled ABOVE the wait and read registers the data has not reached yet
    asm volatile("" : "+v"(cbv[0]), "+v"(cbv[1]));
    f32x4 v = (cbv[0] + cbv[1]) * pre;
    *reinterpret_cast<f32x4 *>(&cbs[wave][4 * lane]) = v;
  };
  // accumulators <- parked values: lane (row, half) holds channels 8 q + 4 half .. + 3 of block b
  auto start_acc = [&]() {
#pragma unroll
    for (int b = 0; b < 8; ++b)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(&cbs[wave][32 * b + 8 * q + 4 * half]);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[b][4 * q + e] = v[e];
      }
  };

  int rb = (k0 * mul + add) * 8 + wave;
  xp = g.Af + (size_t)rb * g.af_stride + lane16;
  load_k0(0);
  load_k1(0);
  xp += FRAG_BLOCK_BYTES;
  load_k0(1);
  load_k1(1);
  xp += FRAG_BLOCK_BYTES;                 // -> piece 2
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) dma(p * R_PIECE_BYTES, p, jj);
  load_cb(rb);
  wait_vm<0>();
  park_cb();
  start_acc();
  __builtin_amdgcn_s_barrier();

  // W fragments of the FIRST block pair of a k step are fetched one k step ahead (in the last block pair's slot of the
  // previous k step), so no k step starts by waiting for LDS; across pieces that needs the piece barrier BEFORE the
  // last block pair of k step 1 -- legal: by then a wave has issued (and, after lgkmcnt(0), completed) every read of
  // this piece's slot, and the transfers of piece p + 1 were waited for at the end of k step 0.
  half8 c[2][2];
  {
    const half8 *w0 = reinterpret_cast<const half8 *>(smem) + lane;
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      c[o][0] = w0[(2 * o) * 64];
      c[o][1] = w0[(2 * o + 1) * 64];
    }
  }
  const unsigned wbytes = (unsigned)np * R_PIECE_BYTES;
  unsigned doff = 3 * R_PIECE_BYTES;
  for (int k = k0; k < kend; k += kstep) {
    const int m0 = rb * 32;
    // the row block whose first pieces the END of this tile's loop fetches (the last tile re-reads its own)
    const int rb_next = k + kstep < kend ? ((k + kstep) * mul + add) * 8 + wave : rb;
    const unsigned char *next_base = g.Af + (size_t)rb_next * g.af_stride + lane16;
    for (int p4 = 0; p4 < np; p4 += 4) {
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) {
        const int xs = ps & 1;
        const half8 *w = reinterpret_cast<const half8 *>(smem + ps * R_PIECE_BYTES) + lane;
        const half8 *wn = reinterpret_cast<const half8 *>(smem + ((ps + 1) & 3) * R_PIECE_BYTES) + lane;
        if (p4 == 0) {
          // the next tile's start values: fetched with piece 0 (16 loads will follow by the end of piece 1), parked in
          // LDS at piece 2 -- after this tile's own start values were read (start_acc above / at the end of the loop)
          if (ps == 0) load_cb(rb_next);
          if (ps == 2) park_cb();
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          half8 nx[2][2];
          const half8 h0 = __builtin_bit_cast(half8, F[xs][2 * s]);
          const half8 l0 = __builtin_bit_cast(half8, F[xs][2 * s + 1]);
#pragma unroll
          for (int bp = 0; bp < 4; ++bp) {
            if (bp < 3) {
#pragma unroll
              for (int o = 0; o < 2; ++o) {
                nx[o][0] = w[(s * 16 + 4 * bp + 4 + 2 * o) * 64];
                nx[o][1] = w[(s * 16 + 4 * bp + 5 + 2 * o) * 64];
              }
            } else if (s == 0) {
#pragma unroll
              for (int o = 0; o < 2; ++o) {
                nx[o][0] = w[(16 + 2 * o) * 64];
                nx[o][1] = w[(16 + 2 * o + 1) * 64];
              }
            } else {
              asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
              __builtin_amdgcn_s_barrier();
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int o = 0; o < 2; ++o) {
                nx[o][0] = wn[(2 * o) * 64];
                nx[o][1] = wn[(2 * o + 1) * 64];
              }
            }
            const int b0 = 2 * bp, b1 = 2 * bp + 1;
            acc[b0] = mfma(c[0][0], h0, acc[b0]);
            acc[b1] = mfma(c[1][0], h0, acc[b1]);
            acc[b0] = mfma(c[0][0], l0, acc[b0]);
            acc[b1] = mfma(c[1][0], l0, acc[b1]);
            acc[b0] = mfma(c[0][1], h0, acc[b0]);
            acc[b1] = mfma(c[1][1], h0, acc[b1]);
            if (!(bp & 1)) dma(doff, (ps + 3) & 3, 2 * s + (bp >> 1));
#pragma unroll
            for (int q = 0; q < 6; ++q) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              if (q < 4) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
              if (q == 3 && !(bp & 1)) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int o = 0; o < 2; ++o) {
              c[o][0] = nx[o][0];
              c[o][1] = nx[o][1];
            }
          }
          // this k step's fragments are consumed (the matrix instructions above have read them): reload the
          // registers with the same k step of piece p + 2 (of the next tile's first pieces near the end)
          if (s == 0) load_k0(xs);
          else load_k1(xs);
          wait_vm<12>();
          // The loads are opaque to the compiler: it believes F was written when the asm was issued and will move
          // the next k step's first matrix instruction ABOVE the wait (tools/audit_vmcnt.py caught exactly that in
          // the first build).  Volatile asm statements keep their order, so these empty ones stay behind the wait,
          // and the fragments the next k step reads now depend on them.
          if (s == 0) asm volatile("" : "+v"(F[xs][2]), "+v"(F[xs][3]));
          else asm volatile("" : "+v"(F[xs ^ 1][0]), "+v"(F[xs ^ 1][1]));
          __builtin_amdgcn_sched_barrier(0);
        }
        // piece p + 3 is what the loads issued during piece p + 1 fetch: past this tile's end, the next tile's piece 0
        xp = (ps == 1 && p4 + 4 == np) ? next_base : xp + FRAG_BLOCK_BYTES;
        doff = doff + R_PIECE_BYTES < wbytes ? doff + R_PIECE_BYTES : 0;
      }
    }
    // ---- tile boundary: the operand streams of the next tile are already in flight; only the accumulators turn over
    if (STORE) frag_epilogue(g, acc, rb, lane, n0);
    if (g.pool) {
      Args gz = g;                    // bias + group bias are already in the accumulators
      gz.bias = g.zeros;
      gz.gbias = g.zeros;
      gz.gbias_stride = 0;
      pool_epilogue(gz, reinterpret_cast<unsigned char *>(red), acc, wave, lane, m0, n0);
    }
    start_acc();
    rb = rb_next;
  }
  wait_vm<0>();                       // the last tile's look-ahead transfers: nothing may be in flight at s_endpgm
}

}  // namespace

// two layouts: the 128 x 128 tile stream, then (for N % 256 == 0, K % 128 == 0) the row-owner stream
// three layouts: tile stream, row-owner stream (fp32 rows), row-owner stream in frag k order
RFD_API size_t rfd_gemm_packed_bytes(int N, int K) { return (size_t)N * K * 2 * sizeof(_Float16) * 3; }

// W [N][K] fp32 (device) -> packed (device).  N % 128 == 0, K % 32 == 0.
RFD_API int rfd_gemm_pack_w(int N, int K, int sw, const float *W, void *packed, void *stream) {
  if (N % BN || K % BK) { rfd_set_error("rfd_gemm_pack_w: N % 128 or K % 32", hipErrorInvalidValue); return (int)hipErrorInvalidValue; }
  const size_t total = (size_t)N * K * 2;
  hipLaunchKernelGGL(gemm_pack_w_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, N, K, sw, W, (_Float16 *)packed);
  RFD_CHECK_LAUNCH();
  if (N % RN == 0 && K % 128 == 0) {
    hipLaunchKernelGGL(gemm_pack_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, N, K, sw, W, (_Float16 *)packed + total);
    RFD_CHECK_LAUNCH();
    hipLaunchKernelGGL(gemm_pack_frag_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, N, K, sw, W, (_Float16 *)packed + 2 * total);
    RFD_CHECK_LAUNCH();
  }
  return 0;
}

// C = act(A) W^T (+bias)(+gbias[m / rows_per_group])(+R), optional ReLU in/out.
// M % 128 == 0, N % 128 == 0, K % 32 == 0; lda/ldc/ldr in floats, lda % 4 == 0,
// A 16-byte aligned.  sa / sw: power-of-two operand scalings (sw as packed).
RFD_API int rfd_gemm_f16x3(int M, int N, int K, const float *A, int lda, const void *packed_w,
                           float *C, int ldc, const float *bias, const float *gbias,
                           int rows_per_group, const float *R, int ldr, int relu_in, int relu_out,
                           int sa, int sw, float *pool_max, int pool_signed, void *stream) {
  if (M <= 0) return 0;
  if (!C && !pool_max) {
    rfd_set_error("rfd_gemm_f16x3: C == NULL without pool_max", hipErrorInvalidValue);
    return (int)hipErrorInvalidValue;
  }
  if (M % BM || N % BN || K % BK || (lda & 3)) {
    rfd_set_error("rfd_gemm_f16x3: shape not a multiple of the 128x128x32 tile", hipErrorInvalidValue);
    return (int)hipErrorInvalidValue;
  }
  Args g;
  g.A = A; g.lda = lda; g.Wp = (const half8 *)packed_w; g.C = C; g.ldc = ldc; g.bias = bias;
  g.gbias = gbias; g.rows_per_group = rows_per_group > 0 ? rows_per_group : 1; g.R = R; g.ldr = ldr;
  g.M = M; g.N = N; g.K = K; g.relu_in = relu_in; g.relu_out = relu_out;
  g.a_scale = ldexpf(1.f, sa); g.out_scale = ldexpf(1.f, -(sa + sw));
  g.pool = pool_max;
  g.pool_signed = pool_signed;
  g.gbias_stride = N;
  RfdWorkspace *ws0;
  {
    int rc0 = rfd_get_workspace(&ws0);
    if (rc0) return rc0;
  }
  g.status = rfd_status_word(ws0, (hipStream_t)stream);
  // RFD_GEMM_TILE_ONLY: A/B switch of tools/gemm_bench.py (the tile kernel for every shape), read once per process
  static const bool tile_only = getenv("RFD_GEMM_TILE_ONLY") != nullptr;
  const bool aligned = !(ldc & 3) && !(ldr & 3) && !((uintptr_t)C & 15) && !((uintptr_t)R & 15) &&
                       !((uintptr_t)bias & 15) && !((uintptr_t)gbias & 15);
  if (M % RM == 0 && N % RN == 0 && N <= RFD_ZEROS_FLOATS && K % 128 == 0 && aligned &&
      ((!gbias && !pool_max) || g.rows_per_group % 64 == 0) && !tile_only) {
    RfdWorkspace *ws;
    int rc = rfd_get_workspace(&ws);
    if (rc) return rc;
    if (!g.bias) g.bias = ws->zeros;
    if (!g.gbias) {
      g.gbias = ws->zeros;       // every group reads the same zero vector
      g.gbias_stride = 0;
    }
    const dim3 grid((M / RM) * (N / RN));
    hipStream_t s = (hipStream_t)stream;
    if (relu_in && R) hipLaunchKernelGGL((gemm_rows8_kernel<true, true>), grid, dim3(512), 0, s, g);
    else if (relu_in) hipLaunchKernelGGL((gemm_rows8_kernel<true, false>), grid, dim3(512), 0, s, g);
    else if (R) hipLaunchKernelGGL((gemm_rows8_kernel<false, true>), grid, dim3(512), 0, s, g);
    else hipLaunchKernelGGL((gemm_rows8_kernel<false, false>), grid, dim3(512), 0, s, g);
  } else {
    if (pool_max || !C) {
      rfd_set_error("rfd_gemm_f16x3: pool_max / C == NULL need the row-owner kernel (M, N % 256, K % 128, "
                    "rows_per_group % 64, 16-byte aligned operands)", hipErrorInvalidValue);
      return (int)hipErrorInvalidValue;
    }
    hipLaunchKernelGGL(gemm_f16x3_kernel, dim3((M / BM) * (N / BN)), dim3(256), 0, (hipStream_t)stream, g);
  }
  RFD_CHECK_LAUNCH();
  return 0;
}

// ---- frag rows (see "Fragment-ordered split activations" above) ------------------------------------------------
// bytes of a frag buffer of M rows x C channels (M % 32 == 0, C % 32 == 0): [M/32][C/32][4096]
RFD_API size_t rfd_frag_bytes(int M, int C) { return (size_t)(M / 32) * (C / 32) * FRAG_BLOCK_BYTES; }

// x [M][ldx] fp32 -> frag rows of relu?(x) 2^sa.  out = first block of row block 0, rb_stride = bytes between row blocks.
RFD_API int rfd_rows_to_frag(int M, int C, const float *x, int ldx, int relu, int sa, void *out, long rb_stride,
                             void *stream) {
  if (M <= 0 || C <= 0) return 0;
  if (M % 32 || C % 32 || (ldx & 3) || ((uintptr_t)x & 15) || ((uintptr_t)out & 15) || (rb_stride & 15)) {
    rfd_set_error("rfd_rows_to_frag: M % 32, C % 32, 16-byte aligned rows / blocks", hipErrorInvalidValue);
    return (int)hipErrorInvalidValue;
  }
  RfdWorkspace *ws;
  int rc = rfd_get_workspace(&ws);
  if (rc) return rc;
  const size_t units = (size_t)(M / 32) * (C / 32);
  hipLaunchKernelGGL(rows_to_frag_kernel, dim3((unsigned)((units + 3) / 4)), dim3(256), 0, (hipStream_t)stream, M, C, x,
                     ldx, relu, ldexpf(1.f, sa), (unsigned char *)out, rb_stride, rfd_status_word(ws, (hipStream_t)stream));
  RFD_CHECK_LAUNCH();
  return 0;
}

RFD_API int rfd_frag_to_rows(int M, int C, const void *in, long rb_stride, int sa, float *x, int ldx, void *stream) {
  if (M <= 0 || C <= 0) return 0;
  if (M % 32 || C % 32 || (ldx & 3) || ((uintptr_t)x & 15) || ((uintptr_t)in & 15) || (rb_stride & 15)) {
    rfd_set_error("rfd_frag_to_rows: M % 32, C % 32, 16-byte aligned rows / blocks", hipErrorInvalidValue);
    return (int)hipErrorInvalidValue;
  }
  const size_t units = (size_t)(M / 32) * (C / 32);
  hipLaunchKernelGGL(frag_to_rows_kernel, dim3((unsigned)((units + 3) / 4)), dim3(256), 0, (hipStream_t)stream, M, C,
                     (const unsigned char *)in, rb_stride, ldexpf(1.f, -sa), x, ldx);
  RFD_CHECK_LAUNCH();
  return 0;
}

// gbias_stride: floats between the groups' rows of gbias (0 = N: contiguous; a column window of a wider matrix otherwise).
// C_frag = split(relu(A W^T + bias + gbias[m / rows_per_group]) 2^sa) with A given as frag rows (already rectified and
// scaled by 2^sa by ITS producer).  M % 256 == 0, N % 256 == 0, K % 128 == 0, rows_per_group % 64 == 0 when gbias or
// pool_max is given.  C_frag may be NULL with pool_max (max over the rows of each group of the fp32 result: max(0, .)
// into a zero-initialised pool, or the plain max into a -inf-initialised one with pool_signed).
RFD_API int rfd_gemm_f16x3_frag(int M, int N, int K, const void *A_frag, long a_rb_stride, const void *packed_w,
                                void *C_frag, long c_rb_stride, const float *bias, const float *gbias,
                                int gbias_stride, int rows_per_group, int sa, int sw, float *pool_max, int pool_signed,
                                void *stream) {
  if (M <= 0) return 0;
  if (!C_frag && !pool_max) {
    rfd_set_error("rfd_gemm_f16x3_frag: C_frag == NULL without pool_max", hipErrorInvalidValue);
    return (int)hipErrorInvalidValue;
  }
  const int rpg = rows_per_group > 0 ? rows_per_group : 1;
  if (M % RM || N % RN || K % 128 || N > RFD_ZEROS_FLOATS || ((gbias || pool_max) && rpg % 64) || (a_rb_stride & 15) ||
      (c_rb_stride & 15) || ((uintptr_t)A_frag & 15) || ((uintptr_t)C_frag & 15) || ((uintptr_t)bias & 15) ||
      ((uintptr_t)gbias & 15) || (gbias_stride & 3)) {
    rfd_set_error("rfd_gemm_f16x3_frag: need M % 256, N % 256, K % 128, rows_per_group % 64, 16-byte aligned operands",
                  hipErrorInvalidValue);
    return (int)hipErrorInvalidValue;
  }
  RfdWorkspace *ws;
  int rc = rfd_get_workspace(&ws);
  if (rc) return rc;
  Args g;
  g.A = nullptr; g.lda = 0; g.Wp = (const half8 *)packed_w; g.C = nullptr; g.ldc = 0;
  g.bias = bias ? bias : ws->zeros;
  g.gbias = gbias ? gbias : ws->zeros; g.gbias_stride = gbias ? N : 0;
  g.rows_per_group = (gbias || pool_max) ? rpg : M;      // one group when nothing is per group
  g.R = nullptr; g.ldr = 0; g.M = M; g.N = N; g.K = K; g.relu_in = 0; g.relu_out = 0;
  g.a_scale = ldexpf(1.f, sa); g.out_scale = ldexpf(1.f, -(sa + sw));
  g.pool = pool_max; g.pool_signed = pool_signed;
  g.status = rfd_status_word(ws, (hipStream_t)stream);
  g.Af = (const unsigned char *)A_frag; g.af_stride = a_rb_stride;
  g.Cf = (unsigned char *)C_frag; g.cf_stride = c_rb_stride;
  g.zeros = ws->zeros;
  g.cb1 = gbias; g.cb1_stride = gbias ? (gbias_stride > 0 ? gbias_stride : N) : 0;
  g.cb2 = bias; g.cb2_stride = 0;
  // persistent: one workgroup per CU (144 KiB of LDS each), a multiple of the n tiles so that a workgroup keeps ONE W
  // stream, and of 8 x n tiles when possible (XCD-aware tile lists, see the kernel)
  const int ntiles = N / RN, tiles = (M / RM) * ntiles;
  int nwg = ws->num_cu / (8 * ntiles) * (8 * ntiles);
  if (nwg == 0) nwg = ws->num_cu / ntiles * ntiles;
  if (nwg == 0) nwg = ntiles;
  if (nwg > tiles) nwg = tiles;
  // cb2 is always dereferenced (one code path): zeros when absent
  if (!g.cb1) { g.cb1 = ws->zeros; g.cb1_stride = 0; }
  if (!g.cb2) { g.cb2 = ws->zeros; g.cb2_stride = 0; }
  const dim3 grid(nwg);
  if (C_frag) hipLaunchKernelGGL((gemm_rowsf_kernel<true>), grid, dim3(512), 0, (hipStream_t)stream, g);
  else hipLaunchKernelGGL((gemm_rowsf_kernel<false>), grid, dim3(512), 0, (hipStream_t)stream, g);
  RFD_CHECK_LAUNCH();
  return 0;
}
