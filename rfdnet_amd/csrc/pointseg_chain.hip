// pointseg_chain.hip -- the three PointNet feature chains of the skip-propagation PointSeg / STN nets, each as ONE kernel:
//     x (M, d)  ->  [d -> 64]  ->  64 -> 128  ->  128 -> 1024  ->  max over the P points of a proposal
// (models/iscnet/modules/pointseg.py:7-42 STN3d, :45-79 STNkd, :82-129 PointNetEncoder: conv1/conv2/conv3 + folded
// BatchNorm + ReLU, torch.max(x, 2)).  Before: per chain two or three GEMM launches writing / re-reading the 64- and
// 128-wide intermediates (200 MB each way) and a 128 -> 1024 pool-only GEMM whose max over the points was a cross-lane
// reduction of every accumulator register (190 TFLOP/s).  Here the intermediates stay in registers (the accumulator
// layout of one layer is the B-fragment layout of the next, as in the occupancy decoder) and the last layer is issued
// with the operands swapped -- A = activations, B = weights -- so an accumulator holds ONE output channel per lane and
// four POINTS per register: the max over the points is 15 v_max per lane and two lane exchanges per 16-channel tile.
//
// Arithmetic = the split-precision GEMM's (csrc/gemm_f16x3.hip): f16 (hi, lo) splits of both operands, three
// v_mfma_f32_16x16x32_f16 per product, fp32 accumulate, activations scaled by 2^sa, weights by 2^sw (exact).
// One workgroup = one proposal (P = 1024 points): a wave owns P/8 points, 64 at a time (four 16-point groups share every
// weight fragment read); the 128 -> 1024 weights (512 KiB of fragments) stream through a 3-slot LDS ring by LDS-DMA,
// the two small layers' weights sit in LDS for the kernel's lifetime.  MODE: 1 = first layer d <= 8 on the VALU (STN3d),
// 2 = first layer 64 -> 64 on the matrix cores (STNkd), 0 = no first layer (PointNetEncoder conv2 / conv3).
#include "common.h"
#include "../../include/rfd_occ.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

constexpr int C1 = 64, C2 = 128, C3 = 1024;           // C3: the LARGEST last-layer width (LDS is sized for it)
constexpr int PIECE = 32 * 1024;                       // 4 output tiles of the last layer x (4 k-steps x hi, lo) x 1 KiB
constexpr int N_PIECES = C3 / 16 / 4;                  // 16 at the largest width
constexpr int W3_BYTES = N_PIECES * PIECE;             // 512 KiB at the largest width
// the last layer's width c3 is a run-time multiple of 64 (one 32-KiB piece per 64 channels): 1024 for the PointNet
// encoders of pointseg.py, 256 for STN_Group's STN3d (pointnet2_modules.py:420-466)
__host__ __device__ inline int chain_pieces(int c3) { return c3 / 64; }
constexpr int W2_BYTES = (C2 / 16) * 2 * 2 * 1024;     // 8 tiles x 2 k-steps x (hi, lo): 32 KiB
constexpr int W1_BYTES = (C1 / 16) * 2 * 2 * 1024;     // 4 tiles x 2 k-steps x (hi, lo): 16 KiB
constexpr int OFF_W2 = 3 * PIECE, OFF_W1 = OFF_W2 + W2_BYTES, OFF_B = OFF_W1 + W1_BYTES;
constexpr int B_FLOATS = C1 + C2 + C3 + C1 * 8 + C3;   // b1, b2, b3, raw W1 (64 x 8), pooled maxima
constexpr int SMEM = OFF_B + B_FLOATS * 4;             // 154 368 B

__device__ __forceinline__ f32x4 mfma16(half8 a, half8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// k order of a 32-wide k-step whose operand is a previous layer's accumulators: lane group kg, slot j
__host__ __device__ inline int chain_k(int ks, int kg, int j) { return 32 * ks + 16 * (j >> 2) + 4 * kg + (j & 3); }

// packed = [W3 stream c3 / 64 x 32 KiB][W2 32 KiB][W1 16 KiB]; every fragment = 64 lanes x 8 halves
__global__ void chain_pack_kernel(int mode, int c3, const float *__restrict__ W1, const float *__restrict__ W2,
                                  const float *__restrict__ W3, int sw1, int sw2, int sw3, _Float16 *__restrict__ packed) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t W3_BYTES = (size_t)chain_pieces(c3) * PIECE;          // (shadows the largest-width constant)
  const size_t total = (size_t)(W3_BYTES + W2_BYTES + W1_BYTES) / 2;
  if (e >= total) return;
  const int j = e & 7, lane = (e >> 3) & 63, idx = lane & 15, kg = lane >> 4;
  size_t frag = e >> 9;
  float w;
  int split;
  if (frag < (size_t)W3_BYTES / 1024) {                 // B operand of the last layer: column idx = output channel
    split = frag & 1;
    const int ks = (frag >> 1) & 3, tile = (int)(frag >> 3);
    w = ldexpf(W3[(size_t)(16 * tile + idx) * C2 + chain_k(ks, kg, j)], sw3);
  } else if ((frag -= W3_BYTES / 1024) < (size_t)W2_BYTES / 1024) {   // A operand: row idx = output channel
    split = frag & 1;
    const int ks = (frag >> 1) & 1, tile = (int)(frag >> 2);
    const int k = mode == 0 ? 32 * ks + 8 * kg + j : chain_k(ks, kg, j);   // input straight from memory: natural order
    w = ldexpf(W2[(size_t)(16 * tile + idx) * C1 + k], sw2);
  } else {
    frag -= W2_BYTES / 1024;
    split = frag & 1;
    const int ks = (frag >> 1) & 1, tile = (int)(frag >> 2);
    w = mode == 2 ? ldexpf(W1[(size_t)(16 * tile + idx) * C1 + 32 * ks + 8 * kg + j], sw1) : 0.f;
  }
  const _Float16 hi = (_Float16)w;
  const _Float16 lo = (_Float16)(w - (float)hi);
  packed[e] = split == 0 ? hi : lo;
}

__device__ __forceinline__ unsigned pk_max_u16(unsigned a, unsigned b) {
  unsigned r;
  asm("v_pk_max_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// two scaled values -> packed f16 hi (round to zero) and lo words
__device__ __forceinline__ void split2(float a0, float a1, unsigned &hiw, unsigned &low, unsigned &amax16) {
  hiw = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a0, a1));
  amax16 = pk_max_u16(amax16, hiw & 0x7fff7fffu);
  float r0, r1;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hiw), "v"(a0));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hiw), "v"(a1));
  low = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(r0, r1));
}

// accumulators of channel tiles 2ks, 2ks+1 (+ bias, ReLU, scale) -> the next layer's operand fragment pair of k-step ks
__device__ __forceinline__ void act_pair(const f32x4 &x0, const f32x4 &x1, const float *bias, int ch, float oscale,
                                         float ascale, half8 &hi, half8 &lo, unsigned &amax16) {
  const f32x4 b0 = *reinterpret_cast<const f32x4 *>(bias + ch), b1 = *reinterpret_cast<const f32x4 *>(bias + ch + 16);
  float v[8];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float a = __builtin_fmaf(x0[r], oscale, b0[r]), c = __builtin_fmaf(x1[r], oscale, b1[r]);
    v[r] = (a > 0.f ? a : 0.f) * ascale;
    v[4 + r] = (c > 0.f ? c : 0.f) * ascale;
  }
  unsigned hw[4], lw[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) split2(v[2 * q], v[2 * q + 1], hw[q], lw[q], amax16);
  hi = __builtin_bit_cast(half8, u32x4{hw[0], hw[1], hw[2], hw[3]});
  lo = __builtin_bit_cast(half8, u32x4{lw[0], lw[1], lw[2], lw[3]});
}

struct ChainArgs {
  int M, P, d_in, ldx, relu3, c3;
  const float *x;
  const half8 *packed;
  const float *W1raw, *b1, *b2, *b3;
  float ascale, os1, os2, os3;      // 2^sa, 2^-(sa+sw1), 2^-(sa+sw2), 2^-(sa+sw3)
  float *out;
  unsigned *status;
};

__device__ __forceinline__ void pool_lds(float *p, float v) {          // running max in LDS, any sign (initialised to -inf)
  if (v >= 0.f) atomicMax(reinterpret_cast<int *>(p), __float_as_int(v + 0.f));
  else atomicMin(reinterpret_cast<unsigned *>(p), __float_as_uint(v));
}

template <int MODE>
__global__ __launch_bounds__(512) void chain_kernel(ChainArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];
  float *s_b1 = reinterpret_cast<float *>(smem + OFF_B), *s_b2 = s_b1 + C1, *s_b3 = s_b2 + C2;
  float *s_w1 = s_b3 + C3, *s_red = s_w1 + C1 * 8;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int n = lane & 15, g = lane >> 4, g4 = 4 * g;
  const int prop = blockIdx.x;
  const char *gp = reinterpret_cast<const char *>(a.packed);
  unsigned amax16 = 0u;
  const int c3 = a.c3, n_pieces = chain_pieces(c3);     // run-time last-layer width

  // ---- per-workgroup constants: the two small layers' fragments, biases, pooled maxima
  {
    const u32x4 *src = reinterpret_cast<const u32x4 *>(gp + (size_t)n_pieces * PIECE);
    u32x4 *dst = reinterpret_cast<u32x4 *>(smem + OFF_W2);
    for (int i = t; i < (W2_BYTES + (MODE == 2 ? W1_BYTES : 0)) / 16; i += 512) dst[i] = src[i];
    for (int i = t; i < C1; i += 512) s_b1[i] = MODE ? a.b1[i] : 0.f;
    for (int i = t; i < C2; i += 512) s_b2[i] = a.b2[i];
    for (int i = t; i < c3; i += 512) {
      s_b3[i] = a.b3[i];
      s_red[i] = -__builtin_inff();
    }
    if (MODE == 1)
      for (int i = t; i < C1 * 8; i += 512) s_w1[i] = (i & 7) < a.d_in ? a.W1raw[(i >> 3) * a.d_in + (i & 7)] : 0.f;
  }
  const int pts_per_wave = a.P / 8, passes = pts_per_wave / 64, total_pieces = passes * n_pieces;
  auto dma_piece = [&](int p) {          // wave w moves fragments 4w .. 4w+3 of piece p (32 fragments)
    const char *src = gp + (size_t)(p % n_pieces) * PIECE + (wave * 4) * 1024 + lane * 16;
    unsigned char *dst = smem + (p % 3) * PIECE + (wave * 4) * 1024;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      __builtin_amdgcn_global_load_lds((gbl_void *)(src + q * 1024), (lds_void *)(dst + q * 1024), 16, 0, 0);
  };
  dma_piece(0);
  dma_piece(1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const half8 *w2 = reinterpret_cast<const half8 *>(smem + OFF_W2) + lane;
  const half8 *w1 = reinterpret_cast<const half8 *>(smem + OFF_W1) + lane;
  for (int pass = 0; pass < passes; ++pass) {
    const size_t row0 = (size_t)prop * a.P + (size_t)wave * pts_per_wave + (size_t)pass * 64;
    half8 h2hi[4][4], h2lo[4][4];                                     // [group][k-step]: the last layer's A operands
#pragma unroll
    for (int grp = 0; grp < 4; ++grp) {
      const float *xr = a.x + (row0 + 16 * grp + n) * (size_t)a.ldx;
      half8 h1hi[2], h1lo[2];
      if (MODE == 1) {
        // first layer on the VALU, in the accumulator layout: lane (point n, g) computes channels 16 tt + 4 g + r
        float xin[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) xin[j] = j < a.d_in ? xr[j] : 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          float v[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const int ch = 32 * ks + 16 * (q >> 2) + g4 + (q & 3);
            float acc = s_b1[ch];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc = __builtin_fmaf(s_w1[ch * 8 + j], xin[j], acc);
            v[q] = (acc > 0.f ? acc : 0.f) * a.ascale;
          }
          unsigned hw[4], lw[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) split2(v[2 * q], v[2 * q + 1], hw[q], lw[q], amax16);
          h1hi[ks] = __builtin_bit_cast(half8, u32x4{hw[0], hw[1], hw[2], hw[3]});
          h1lo[ks] = __builtin_bit_cast(half8, u32x4{lw[0], lw[1], lw[2], lw[3]});
        }
      } else {
        // 64 input channels from memory as fragments in their natural k order: lane (n, g) holds k = 32 ks + 8 g + j
        half8 inhi[2], inlo[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const f32x4 u0 = *reinterpret_cast<const f32x4 *>(xr + 32 * ks + 8 * g);
          const f32x4 u1 = *reinterpret_cast<const f32x4 *>(xr + 32 * ks + 8 * g + 4);
          unsigned hw[4], lw[4];
          split2(u0[0] * a.ascale, u0[1] * a.ascale, hw[0], lw[0], amax16);
          split2(u0[2] * a.ascale, u0[3] * a.ascale, hw[1], lw[1], amax16);
          split2(u1[0] * a.ascale, u1[1] * a.ascale, hw[2], lw[2], amax16);
          split2(u1[2] * a.ascale, u1[3] * a.ascale, hw[3], lw[3], amax16);
          inhi[ks] = __builtin_bit_cast(half8, u32x4{hw[0], hw[1], hw[2], hw[3]});
          inlo[ks] = __builtin_bit_cast(half8, u32x4{lw[0], lw[1], lw[2], lw[3]});
        }
        if (MODE == 2) {
          f32x4 d1[4];
#pragma unroll
          for (int tt = 0; tt < 4; ++tt) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
              const half8 wh = w1[((tt * 2 + ks) * 2) * 64], wl = w1[((tt * 2 + ks) * 2 + 1) * 64];
              acc = mfma16(wh, inhi[ks], acc);
              acc = mfma16(wh, inlo[ks], acc);
              acc = mfma16(wl, inhi[ks], acc);
            }
            d1[tt] = acc;
          }
#pragma unroll
          for (int ks = 0; ks < 2; ++ks)
            act_pair(d1[2 * ks], d1[2 * ks + 1], s_b1, 32 * ks + g4, a.os1, a.ascale, h1hi[ks], h1lo[ks], amax16);
        } else {
          h1hi[0] = inhi[0]; h1hi[1] = inhi[1];
          h1lo[0] = inlo[0]; h1lo[1] = inlo[1];
        }
      }
      // MODE 1: the first layer above is a VALU prologue on LDS-loaded weights (ds_read -> v_mov / v_pk_fma, ~130
      // consumers) running straight into MFMA code -- the shape of the decoder's tile prologue, the one place where round
      // 2's static-priority build lost an LDS-loaded value while the SIMD partner was already in its MFMA code
      // (profiles/r03_fault_model.txt).  Same belt as there: all eight waves finish it before any enters the MFMAs
      // (tools/audit_prologue_lds.py finds the shape; tests/test_isa_audit.py asserts it is gone).  Uniform control flow.
      if (MODE == 1) __syncthreads();
      // ---- 64 -> 128
      __builtin_amdgcn_sched_barrier(0);
      f32x4 d2[8];
#pragma unroll
      for (int tt = 0; tt < 8; ++tt) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const half8 wh = w2[((tt * 2 + ks) * 2) * 64], wl = w2[((tt * 2 + ks) * 2 + 1) * 64];
          acc = mfma16(wh, h1hi[ks], acc);
          acc = mfma16(wh, h1lo[ks], acc);
          acc = mfma16(wl, h1hi[ks], acc);
        }
        d2[tt] = acc;
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        act_pair(d2[2 * ks], d2[2 * ks + 1], s_b2, 32 * ks + g4, a.os2, a.ascale, h2hi[grp][ks], h2lo[grp][ks], amax16);
      __builtin_amdgcn_sched_barrier(0);      // one group at a time: four interleaved groups do not fit the register file
    }

    // ---- 128 -> 1024 with the operands swapped (A = activations, B = weights) + max over the wave's 64 points
    for (int i = 0; i < n_pieces; ++i) {
      const int p = pass * n_pieces + i;
      if (p + 2 < total_pieces) dma_piece(p + 2);
      const half8 *w3 = reinterpret_cast<const half8 *>(smem + (p % 3) * PIECE) + lane;
      // fragments of a tile: 4 k-steps x (hi, lo); the next tile's are fetched under this tile's 48 MFMAs into the OTHER
      // register set (the set in use is kept allocated to the end of the tile, nothing is scheduled across tiles: same
      // discipline as the decoder's prefetch, csrc/occ_decoder8.hip)
      half8 cw[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) cw[q] = w3[q * 64];
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        __builtin_amdgcn_sched_barrier(0);
        half8 nw[8];
        if (tt < 3) {
#pragma unroll
          for (int q = 0; q < 8; ++q) nw[q] = w3[((tt + 1) * 8 + q) * 64];
        }
        f32x4 d[4];
#pragma unroll
        for (int grp = 0; grp < 4; ++grp) d[grp] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int grp = 0; grp < 4; ++grp) {
            d[grp] = mfma16(h2hi[grp][ks], cw[2 * ks], d[grp]);
            d[grp] = mfma16(h2lo[grp][ks], cw[2 * ks], d[grp]);
            d[grp] = mfma16(h2hi[grp][ks], cw[2 * ks + 1], d[grp]);
          }
        // lane (n = output channel, g): d[grp][r] = point 16 grp + 4 g + r
        float m = d[0][0];
#pragma unroll
        for (int grp = 0; grp < 4; ++grp)
#pragma unroll
          for (int r = 0; r < 4; ++r) m = d[grp][r] > m ? d[grp][r] : m;
        float o = __shfl_xor(m, 16);
        m = o > m ? o : m;
        o = __shfl_xor(m, 32);
        m = o > m ? o : m;
        if (lane < 16) pool_lds(s_red + 16 * (4 * i + tt) + n, m);
        asm volatile("" ::"v"(cw[0]), "v"(cw[1]), "v"(cw[2]), "v"(cw[3]), "v"(cw[4]), "v"(cw[5]), "v"(cw[6]), "v"(cw[7]));
        if (tt < 3) {
#pragma unroll
          for (int q = 0; q < 8; ++q) cw[q] = nw[q];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      // the piece after this one must have landed (this wave's four transfers of piece p + 2 may still fly)
      if (p + 2 < total_pieces) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }
  // ---- out[prop][c] = relu?(max * 2^-(sa+sw3) + b3[c])   (max commutes with the positive scale, the bias and the ReLU)
  for (int c = t; c < c3; c += 512) {
    float o = __builtin_fmaf(s_red[c], a.os3, s_b3[c]);
    if (a.relu3) o = o > 0.f ? o : 0.f;
    a.out[(size_t)prop * c3 + c] = o;
    // the pooled feature is the next split-precision layer's INPUT (STN fc1, PointSeg's conv1 share): watch what is
    // stored, like the row-owner GEMM's pool path does -- |o| 2^sa beyond f16 would saturate there silently
    if (fabsf(o) * a.ascale >= 65504.f) atomicOr(a.status, 4u);
  }
  if ((amax16 & 0xffffu) >= 0x7bffu || (amax16 >> 16) >= 0x7bffu) atomicOr(a.status, 4u);
}

// =====================================================================================================================
// PointSeg's per-point head (pointseg.py:131-154: conv1 1088 -> 512 on cat([global feature, point feature]), conv2
// 512 -> 256, conv3 256 -> 128, conv4 128 -> k, BatchNorms folded, ReLU between) as ONE kernel.  The 1024 global-feature
// columns of conv1 are one vector per proposal (gbias, computed by the caller), so per point it is
//     64 -> 512 (+ gbias[proposal], ReLU) -> 256 (ReLU) -> 128 (ReLU) -> k = 2 scores.
// Before: four launches writing / re-reading 512-, 256- and 128-wide intermediates of 262 144 rows (1.9 GB).  Here a wave
// owns 16 points at a time and the layers are interleaved the way the occupancy decoder interleaves fc_0 and fc_1: every
// 32-channel slab of the 512-wide layer is rectified in registers and consumed at once as one k-step of the 256-wide layer,
// whose sixteen accumulator tiles are the only wide state (64 registers).  All weights (768 KiB of fragments per pass)
// stream through a 3-slot LDS ring in consumption order; the last layer (128 -> 2) is fp32 on the VALU.
constexpr int HA = 512, HB = 256, HC = 128;
constexpr int HPIECE = 40 * 1024;                 // one slab: 8 fragments of layer a + 32 of layer b (c pieces: 32 + 8 unused)
constexpr int H_SLABS = HA / 32;                  // 16
constexpr int H_CPIECES = (HC / 16) * (HB / 32) * 2 / 32;      // 128 fragments of layer c = 4 pieces
constexpr int H_PIECES = H_SLABS + H_CPIECES;     // 20 per pass
constexpr int H_OFF_T = 3 * HPIECE;
constexpr int H_T_FLOATS = HA + HB + HC + 2 * HC + 8;          // gbias, bias b, bias c, W_d (<= 2 classes), bias d
constexpr int H_SMEM = H_OFF_T + H_T_FLOATS * 4;

// stream = 20 pieces x 40 fragments.  Slab mb: fragments 0..7 = layer a, tiles 2mb, 2mb+1 (A operand, natural k order:
// q = (tt * 2 + ks) * 2 + split); 8..39 = layer b, k-step mb of every output tile (q = 8 + 2 t + split, chain k order).
// Piece 16 + i: fragments 0..31 = layer c, tiles 2i, 2i+1 (q = ((tt * 8 + ks) * 2 + split), chain k order), 32..39 unused.
__global__ void head_pack_kernel(const float *__restrict__ Wa, const float *__restrict__ Wb, const float *__restrict__ Wc,
                                 int swa, int swb, int swc, _Float16 *__restrict__ packed) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)H_PIECES * 40 * 512;
  if (e >= total) return;
  const int j = e & 7, lane = (e >> 3) & 63, idx = lane & 15, kg = lane >> 4;
  const int frag = (int)(e >> 9), piece = frag / 40, q = frag % 40;
  float w = 0.f;
  int split = 0;
  if (piece < H_SLABS) {
    if (q < 8) {
      split = q & 1;
      const int ks = (q >> 1) & 1, tt = q >> 2;
      w = ldexpf(Wa[(size_t)(32 * piece + 16 * tt + idx) * C1 + 32 * ks + 8 * kg + j], swa);
    } else {
      split = q & 1;
      const int t = (q - 8) >> 1;
      w = ldexpf(Wb[(size_t)(16 * t + idx) * HA + chain_k(piece, kg, j)], swb);
    }
  } else if (q < 32) {
    split = q & 1;
    const int ks = (q >> 1) & 7, tt = q >> 4, tile = 2 * (piece - H_SLABS) + tt;
    w = ldexpf(Wc[(size_t)(16 * tile + idx) * HB + chain_k(ks, kg, j)], swc);
  }
  const _Float16 hi = (_Float16)w;
  const _Float16 lo = (_Float16)(w - (float)hi);
  packed[e] = split == 0 ? hi : lo;
}

struct HeadArgs {
  int M, P, ldx, n_cls;
  const float *x;
  const half8 *packed;
  const float *gbias, *bb, *bc, *Wd, *bd;
  float ascale, osa, osb, osc;
  float *out;
  unsigned *status;
};

__global__ __launch_bounds__(512) void head_kernel(HeadArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[H_SMEM];
  float *s_ga = reinterpret_cast<float *>(smem + H_OFF_T), *s_bb = s_ga + HA, *s_bc = s_bb + HB, *s_wd = s_bc + HC;
  float *s_bd = s_wd + 2 * HC;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int n = lane & 15, g = lane >> 4, g4 = 4 * g;
  const int prop = blockIdx.x;
  const char *gp = reinterpret_cast<const char *>(a.packed);
  unsigned amax16 = 0u;
  for (int i = t; i < HA; i += 512) s_ga[i] = a.gbias[(size_t)prop * HA + i];
  for (int i = t; i < HB; i += 512) s_bb[i] = a.bb[i];
  for (int i = t; i < HC; i += 512) s_bc[i] = a.bc[i];
  for (int i = t; i < 2 * HC; i += 512) s_wd[i] = i < a.n_cls * HC ? a.Wd[i] : 0.f;
  if (t < 2) s_bd[t] = t < a.n_cls ? a.bd[t] : 0.f;
  const int passes = a.P / 128, total_pieces = passes * H_PIECES;
  auto dma_piece = [&](int p) {          // wave w moves fragments 5w .. 5w+4 of piece p (40 fragments)
    const char *src = gp + (size_t)(p % H_PIECES) * HPIECE + (wave * 5) * 1024 + lane * 16;
    unsigned char *dst = smem + (p % 3) * HPIECE + (wave * 5) * 1024;
#pragma unroll
    for (int q = 0; q < 5; ++q)
      __builtin_amdgcn_global_load_lds((gbl_void *)(src + q * 1024), (lds_void *)(dst + q * 1024), 16, 0, 0);
  };
  auto end_piece = [&](int p) {          // the piece after p must have landed before anybody reads it
    __builtin_amdgcn_sched_barrier(0);
    if (p + 2 < total_pieces) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };
  dma_piece(0);
  dma_piece(1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int pass = 0; pass < passes; ++pass) {
    const size_t row = (size_t)prop * a.P + (size_t)pass * 128 + wave * 16 + n;
    const float *xr = a.x + row * (size_t)a.ldx;
    half8 inhi[2], inlo[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const f32x4 u0 = *reinterpret_cast<const f32x4 *>(xr + 32 * ks + 8 * g);
      const f32x4 u1 = *reinterpret_cast<const f32x4 *>(xr + 32 * ks + 8 * g + 4);
      unsigned hw[4], lw[4];
      split2(u0[0] * a.ascale, u0[1] * a.ascale, hw[0], lw[0], amax16);
      split2(u0[2] * a.ascale, u0[3] * a.ascale, hw[1], lw[1], amax16);
      split2(u1[0] * a.ascale, u1[1] * a.ascale, hw[2], lw[2], amax16);
      split2(u1[2] * a.ascale, u1[3] * a.ascale, hw[3], lw[3], amax16);
      inhi[ks] = __builtin_bit_cast(half8, u32x4{hw[0], hw[1], hw[2], hw[3]});
      inlo[ks] = __builtin_bit_cast(half8, u32x4{lw[0], lw[1], lw[2], lw[3]});
    }
    f32x4 H2[16];
#pragma unroll
    for (int tt = 0; tt < 16; ++tt) H2[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
    // ---- 16 slabs: layer a tiles (2mb, 2mb+1) -> ReLU -> k-step mb of layer b
    for (int mb = 0; mb < H_SLABS; ++mb) {
      const int p = pass * H_PIECES + mb;
      if (p + 2 < total_pieces) dma_piece(p + 2);
      const half8 *w = reinterpret_cast<const half8 *>(smem + (p % 3) * HPIECE) + lane;
      f32x4 da[2];
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const half8 wh = w[((tt * 2 + ks) * 2) * 64], wl = w[((tt * 2 + ks) * 2 + 1) * 64];
          acc = mfma16(wh, inhi[ks], acc);
          acc = mfma16(wh, inlo[ks], acc);
          acc = mfma16(wl, inhi[ks], acc);
        }
        da[tt] = acc;
      }
      half8 bhi, blo;
      act_pair(da[0], da[1], s_ga, 32 * mb + g4, a.osa, a.ascale, bhi, blo, amax16);
      __builtin_amdgcn_sched_barrier(0);
      half8 cw[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) cw[q] = w[(8 + q) * 64];
#pragma unroll
      for (int tp = 0; tp < 8; ++tp) {          // two output tiles of layer b per step
        __builtin_amdgcn_sched_barrier(0);
        half8 nw[4];
        if (tp < 7) {
#pragma unroll
          for (int q = 0; q < 4; ++q) nw[q] = w[(8 + 4 * (tp + 1) + q) * 64];
        }
        H2[2 * tp] = mfma16(cw[0], bhi, H2[2 * tp]);
        H2[2 * tp + 1] = mfma16(cw[2], bhi, H2[2 * tp + 1]);
        H2[2 * tp] = mfma16(cw[0], blo, H2[2 * tp]);
        H2[2 * tp + 1] = mfma16(cw[2], blo, H2[2 * tp + 1]);
        H2[2 * tp] = mfma16(cw[1], bhi, H2[2 * tp]);
        H2[2 * tp + 1] = mfma16(cw[3], bhi, H2[2 * tp + 1]);
        asm volatile("" ::"v"(cw[0]), "v"(cw[1]), "v"(cw[2]), "v"(cw[3]));
        if (tp < 7) {
#pragma unroll
          for (int q = 0; q < 4; ++q) cw[q] = nw[q];
        }
      }
      end_piece(p);
    }
    // ---- layer b -> ReLU -> layer c (two output tiles per piece), -> ReLU -> scores on the VALU
    half8 chi[8], clo[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
      act_pair(H2[2 * ks], H2[2 * ks + 1], s_bb, 32 * ks + g4, a.osb, a.ascale, chi[ks], clo[ks], amax16);
    // 48 VALU consumers of LDS-read bias rows with no MFMA since the last barrier, layer c's MFMAs right behind: closed
    // with a barrier like the decoder's tile prologue (tools/audit_prologue_lds.py); one per 128-point pass
    __syncthreads();
    float sc0 = 0.f, sc1 = 0.f;
    for (int i = 0; i < H_CPIECES; ++i) {
      const int p = pass * H_PIECES + H_SLABS + i;
      if (p + 2 < total_pieces) dma_piece(p + 2);
      const half8 *w = reinterpret_cast<const half8 *>(smem + (p % 3) * HPIECE) + lane;
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        half8 ch_ = w[(tt * 16) * 64], cl_ = w[(tt * 16 + 1) * 64];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          __builtin_amdgcn_sched_barrier(0);
          half8 nh, nl;
          if (ks < 7) {
            nh = w[(tt * 16 + 2 * ks + 2) * 64];
            nl = w[(tt * 16 + 2 * ks + 3) * 64];
          }
          acc = mfma16(ch_, chi[ks], acc);
          acc = mfma16(ch_, clo[ks], acc);
          acc = mfma16(cl_, chi[ks], acc);
          asm volatile("" ::"v"(ch_), "v"(cl_));
          if (ks < 7) {
            ch_ = nh;
            cl_ = nl;
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        // lane (point n, g): channels 16 tile + 4 g + r of the 128-wide layer -> bias, ReLU, partial scores
        const int ch = 16 * (2 * i + tt) + g4;
        const f32x4 bv = *reinterpret_cast<const f32x4 *>(s_bc + ch);
        const f32x4 w0 = *reinterpret_cast<const f32x4 *>(s_wd + ch), w1 = *reinterpret_cast<const f32x4 *>(s_wd + HC + ch);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = __builtin_fmaf(acc[r], a.osc, bv[r]);
          v = v > 0.f ? v : 0.f;
          sc0 = __builtin_fmaf(w0[r], v, sc0);
          sc1 = __builtin_fmaf(w1[r], v, sc1);
        }
      }
      end_piece(p);
    }
    sc0 += __shfl_xor(sc0, 16);
    sc0 += __shfl_xor(sc0, 32);
    sc1 += __shfl_xor(sc1, 16);
    sc1 += __shfl_xor(sc1, 32);
    if (lane < 16) {
      a.out[row * a.n_cls] = sc0 + s_bd[0];
      if (a.n_cls > 1) a.out[row * a.n_cls + 1] = sc1 + s_bd[1];
    }
  }
  if ((amax16 & 0xffffu) >= 0x7bffu || (amax16 >> 16) >= 0x7bffu) atomicOr(a.status, 4u);
}

}  // namespace

static bool chain_width_ok(int c3) { return c3 >= 64 && c3 <= C3 && c3 % 64 == 0; }

RFD_API size_t rfd_chain_packed_bytes_n(int c3) {
  return chain_width_ok(c3) ? (size_t)chain_pieces(c3) * PIECE + W2_BYTES + W1_BYTES : 0;
}
RFD_API size_t rfd_chain_packed_bytes(void) { return rfd_chain_packed_bytes_n(C3); }

// W1 [64][64] (mode 2) or NULL, W2 [128][64], W3 [c3][128]: fp32, BatchNorm already folded in by the caller.
// c3: the last layer's width, a multiple of 64 up to 1024.
RFD_API int rfd_chain_pack_n(int mode, int c3, const float *W1, const float *W2, const float *W3, int sw1, int sw2,
                             int sw3, void *packed, void *stream) {
  if (mode < 0 || mode > 2 || !W2 || !W3 || (mode == 2 && !W1) || !chain_width_ok(c3)) {
    rfd_set_error("rfd_chain_pack: mode / weights / width (64 .. 1024, multiple of 64)", hipErrorInvalidValue);
    return (int)hipErrorInvalidValue;
  }
  const size_t total = rfd_chain_packed_bytes_n(c3) / 2;
  hipLaunchKernelGGL(chain_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, mode,
                     c3, W1, W2, W3, sw1, sw2, sw3, (_Float16 *)packed);
  RFD_CHECK_LAUNCH();
  return 0;
}
RFD_API int rfd_chain_pack(int mode, const float *W1, const float *W2, const float *W3, int sw1, int sw2, int sw3,
                           void *packed, void *stream) {
  return rfd_chain_pack_n(mode, C3, W1, W2, W3, sw1, sw2, sw3, packed, stream);
}

// x [M][ldx] fp32 rows (d_in <= 8 columns used in mode 1, 64 otherwise; 16-byte aligned rows for modes 0 / 2),
// P points per proposal (P % 512 == 0: 8 waves x 64 points), out [M / P][c3].
RFD_API int rfd_chain_pool_n(int mode, int c3, int M, int P, int d_in, const float *x, int ldx, const void *packed,
                             const float *W1raw, const float *b1, const float *b2, const float *b3, int relu3, int sa,
                             int sw1, int sw2, int sw3, float *out, void *stream) {
  if (M <= 0) return 0;
  if (mode < 0 || mode > 2 || P <= 0 || P % 512 || M % P || (mode == 1 && (d_in < 1 || d_in > 8 || !W1raw)) ||
      (mode != 1 && (d_in != 64 || (ldx & 3) || ((uintptr_t)x & 15))) || (mode && !b1) || !b2 || !b3 || !out ||
      !chain_width_ok(c3)) {
    rfd_set_error("rfd_chain_pool: need P % 512 == 0, M % P == 0, d_in <= 8 (mode 1) or 64 with 16-byte rows, "
                  "last width 64 .. 1024 in steps of 64", hipErrorInvalidValue);
    return (int)hipErrorInvalidValue;
  }
  RfdWorkspace *ws;
  {
    int rc = rfd_get_workspace(&ws);
    if (rc) return rc;
  }
  ChainArgs a;
  a.M = M; a.P = P; a.d_in = d_in; a.ldx = ldx; a.relu3 = relu3; a.c3 = c3; a.x = x; a.packed = (const half8 *)packed;
  a.W1raw = W1raw; a.b1 = b1; a.b2 = b2; a.b3 = b3;
  a.ascale = ldexpf(1.f, sa); a.os1 = ldexpf(1.f, -(sa + sw1)); a.os2 = ldexpf(1.f, -(sa + sw2));
  a.os3 = ldexpf(1.f, -(sa + sw3));
  a.out = out;
  a.status = rfd_status_word(ws, (hipStream_t)stream);
  const dim3 grid(M / P), block(512);
  hipStream_t s = (hipStream_t)stream;
  if (mode == 0) hipLaunchKernelGGL(chain_kernel<0>, grid, block, 0, s, a);
  else if (mode == 1) hipLaunchKernelGGL(chain_kernel<1>, grid, block, 0, s, a);
  else hipLaunchKernelGGL(chain_kernel<2>, grid, block, 0, s, a);
  RFD_CHECK_LAUNCH();
  return 0;
}
RFD_API int rfd_chain_pool(int mode, int M, int P, int d_in, const float *x, int ldx, const void *packed,
                           const float *W1raw, const float *b1, const float *b2, const float *b3, int relu3, int sa,
                           int sw1, int sw2, int sw3, float *out, void *stream) {
  return rfd_chain_pool_n(mode, C3, M, P, d_in, x, ldx, packed, W1raw, b1, b2, b3, relu3, sa, sw1, sw2, sw3, out, stream);
}

RFD_API size_t rfd_head_packed_bytes(void) { return (size_t)H_PIECES * HPIECE; }

// Wa [512][64] (conv1's point-feature columns), Wb [256][512], Wc [128][256]: fp32, BatchNorm folded in by the caller.
RFD_API int rfd_head_pack(const float *Wa, const float *Wb, const float *Wc, int swa, int swb, int swc, void *packed,
                          void *stream) {
  if (!Wa || !Wb || !Wc) {
    rfd_set_error("rfd_head_pack: weights", hipErrorInvalidValue);
    return (int)hipErrorInvalidValue;
  }
  const size_t total = rfd_head_packed_bytes() / 2;
  hipLaunchKernelGGL(head_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, Wa, Wb,
                     Wc, swa, swb, swc, (_Float16 *)packed);
  RFD_CHECK_LAUNCH();
  return 0;
}

// x [M][ldx]: the 64-wide point feature (16-byte aligned rows); gbias [M / P][512] = conv1's bias + its global-feature
// share per proposal; bb [256], bc [128]; Wd [n_cls][128], bd [n_cls] (n_cls <= 2) -> out [M][n_cls] raw scores.
RFD_API int rfd_head_scores(int M, int P, const float *x, int ldx, const void *packed, const float *gbias,
                            const float *bb, const float *bc, const float *Wd, const float *bd, int n_cls, int sa,
                            int swa, int swb, int swc, float *out, void *stream) {
  if (M <= 0) return 0;
  if (P <= 0 || P % 128 || M % P || (ldx & 3) || ((uintptr_t)x & 15) || n_cls < 1 || n_cls > 2 || !gbias || !bb || !bc ||
      !Wd || !bd || !out) {
    rfd_set_error("rfd_head_scores: need P % 128 == 0, M % P == 0, 16-byte rows, n_cls <= 2", hipErrorInvalidValue);
    return (int)hipErrorInvalidValue;
  }
  RfdWorkspace *ws;
  {
    int rc = rfd_get_workspace(&ws);
    if (rc) return rc;
  }
  HeadArgs a;
  a.M = M; a.P = P; a.ldx = ldx; a.n_cls = n_cls; a.x = x; a.packed = (const half8 *)packed; a.gbias = gbias;
  a.bb = bb; a.bc = bc; a.Wd = Wd; a.bd = bd;
  a.ascale = ldexpf(1.f, sa); a.osa = ldexpf(1.f, -(sa + swa)); a.osb = ldexpf(1.f, -(sa + swb));
  a.osc = ldexpf(1.f, -(sa + swc));
  a.out = out;
  a.status = rfd_status_word(ws, (hipStream_t)stream);
  hipLaunchKernelGGL(head_kernel, dim3(M / P), dim3(512), 0, (hipStream_t)stream, a);
  RFD_CHECK_LAUNCH();
  return 0;
}
