// occ_decoder.hip -- fused conditional-batch-norm occupancy decoder (MFMA).
//
// Replaces DecoderCBatchNorm.forward (models/iscnet/modules/occ_decoder.py:
// 110-123) + CResnetBlockConv1d.forward (layers.py:98-107) + CBatchNorm1d
// (layers.py:226-242, eval mode) as called from Generator3D.eval_points
// (generator.py:123-143).  One kernel for the whole 11-layer network.
//
// Decomposition.  A workgroup (4 waves) takes a tile of 128 query points of one
// proposal; each WAVE owns 32 points x all 256 channels.  With
// v_mfma_f32_32x32x16_f16, D[channel, point] = sum_k W[channel, k] * act[k, point]:
// A = weights (streamed), B = activations (registers), D layout: lane holds
// point (lane & 31), channels (r&3) + 8(r>>2) + 4(lane>>5) of a 32-channel
// block.  The B operand of the NEXT layer wants, per lane, 8 consecutive-k
// values of its own point: we define the k-order of every weight matrix as the
// order in which the accumulator layout delivers channels (the pack kernel
// permutes the weight columns), so activations never leave the lane: CBN +
// ReLU + f16 split are applied to accumulator registers and fed straight back
// as B fragments.  No LDS or HBM round trip for activations at all.
//
// Residual stream H' = (h - cumulative fc_1 biases) * 2^KH stays in 8 x 16
// accumulator registers per lane (128 VGPRs) and is the C operand of the
// second GEMM of each block, so "h = h + fc_1(...)" is free.  All biases,
// BN statistics, CBN gamma/beta and power-of-two operand scalings are folded on
// the host into the per-proposal table (see rfdnet_amd/occ_fold.py).
//
// Precision.  Mode F16X3 splits both operands into f16 (hi, lo) and issues
// hi*hi + hi*lo + lo*hi (fp32 accumulate): operands are pre-scaled by powers of
// two (activations 2^6, weights 2^kw) so lo stays a normal f16 over the useful
// range; ~2^-20 relative error per product.  Mode F16X1 issues hi*hi only.
//
// Roofline: 1 312 768 FLOP per query point vs 16 B of HBM traffic => MFMA
// bound.  F16X3 issues 3x the algorithmic MFMA work (peak = 2.5 PF/3).
#include "common.h"
#include "../../include/rfd_occ.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int H = RFD_OCC_HIDDEN;
constexpr int NB = RFD_OCC_BLOCKS;
constexpr int TILE = RFD_OCC_TILE;
constexpr int ROWS = RFD_OCC_TABLE_ROWS;
constexpr int FRAG_HALVES = 64 * 8;           // one A fragment: 64 lanes x 8 f16 = 1 KiB
constexpr int FRAGS_PER_CHUNK = 64;           // (block, mb): 32 GEMM1 + 32 GEMM2 fragments
constexpr size_t PACKED_HALVES = (size_t)NB * 8 * FRAGS_PER_CHUNK * FRAG_HALVES;
constexpr float ACT_SCALE = 64.f;             // 2^ka, ka = 6

// ---- weight packing -----------------------------------------------------------
// Stream order = consumption order.  For block i, output block mb (32 channels
// of fc_0's output == one 32-wide K slab of fc_1's input):
//   fragments  0..31 : fc_0, A[32mb + m][k(ks,h,j)], ks = q>>1, split s = q&1
//   fragments 32..63 : fc_1, A[32ob + m][32mb + 16sub + ...], ob = q>>2,
//                      sub = (q>>1)&1, split s = q&1
// with lane = 32h + m, k(ks,h,j) = 32(ks>>1) + 16(ks&1) + 8(j>>2) + 4h + (j&3).
__global__ void pack_weights_kernel(const float *__restrict__ fc0_w,
                                    const float *__restrict__ fc1_w, int kw0_0,
                                    int kw0_1, int kw0_2, int kw0_3, int kw0_4,
                                    int kw1, _Float16 *__restrict__ packed) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= PACKED_HALVES) return;
  const int j = e & 7;
  const int lane = (e >> 3) & 63;
  const int frag = (int)(e >> 9);
  const int q = frag & 31;
  const int g = (frag >> 5) & 1;
  const int mb = (frag >> 6) & 7;
  const int blk = frag >> 9;
  const int m = lane & 31, h = lane >> 5;
  const int s = q & 1;
  int out_ch, in_ch, kw;
  const float *W;
  if (g == 0) {
    const int ks = q >> 1;
    out_ch = 32 * mb + m;
    in_ch = 32 * (ks >> 1) + 16 * (ks & 1) + 8 * (j >> 2) + 4 * h + (j & 3);
    W = fc0_w + (size_t)blk * H * H;
    kw = blk == 0 ? kw0_0 : blk == 1 ? kw0_1 : blk == 2 ? kw0_2 : blk == 3 ? kw0_3 : kw0_4;
  } else {
    const int ob = q >> 2, sub = (q >> 1) & 1;
    out_ch = 32 * ob + m;
    in_ch = 32 * mb + 16 * sub + 8 * (j >> 2) + 4 * h + (j & 3);
    W = fc1_w + (size_t)blk * H * H;
    kw = kw1;
  }
  const float w = ldexpf(W[(size_t)out_ch * H + in_ch], kw);
  const _Float16 hi = (_Float16)w;  // round-to-nearest
  const _Float16 lo = (_Float16)(w - (float)hi);
  packed[e] = s == 0 ? hi : lo;
}

// ---- helpers --------------------------------------------------------------------
__device__ __forceinline__ f32x16 mfma(half8 a, half8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// relu(s*x + t) of 16 accumulator values -> two B fragments (hi) and (lo).
// Channel of register r: 8(r>>2) + 4*half + (r&3) within the 32-block, so the
// table rows are read as four float4 per 16 registers.
template <bool WITH_LO>
__device__ __forceinline__ void cbn_relu_split(const f32x16 &x, const float *s_row,
                                               const float *t_row, int ch0,
                                               half8 &hi0, half8 &hi1, half8 &lo0,
                                               half8 &lo1, float &amax) {
  float v[16];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 s4 = *reinterpret_cast<const f32x4 *>(s_row + ch0 + 8 * q);
    const f32x4 t4 = *reinterpret_cast<const f32x4 *>(t_row + ch0 + 8 * q);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float a = __builtin_fmaf(s4[e], x[4 * q + e], t4[e]);
      a = a > 0.f ? a : 0.f;
      amax = a > amax ? a : amax;
      v[4 * q + e] = a;
    }
  }
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const float a0 = v[2 * p], a1 = v[2 * p + 1];
    const half2v h2 = __builtin_bit_cast(half2v, __builtin_amdgcn_cvt_pkrtz(a0, a1));
    if (p < 4) { hi0[2 * p] = h2[0]; hi0[2 * p + 1] = h2[1]; }
    else       { hi1[2 * (p - 4)] = h2[0]; hi1[2 * (p - 4) + 1] = h2[1]; }
    if (WITH_LO) {
      const float r0 = a0 - (float)h2[0], r1 = a1 - (float)h2[1];
      const half2v l2 = __builtin_bit_cast(half2v, __builtin_amdgcn_cvt_pkrtz(r0, r1));
      if (p < 4) { lo0[2 * p] = l2[0]; lo0[2 * p + 1] = l2[1]; }
      else       { lo1[2 * (p - 4)] = l2[0]; lo1[2 * (p - 4) + 1] = l2[1]; }
    }
  }
}

template <int TERMS>
__global__ __launch_bounds__(256) void occ_decode_kernel(
    int n_tiles, const float *__restrict__ pts, const int *__restrict__ tile_prop,
    const int *__restrict__ tile_src, const half8 *__restrict__ packed, const float *__restrict__ fc_p_w,
    const float *__restrict__ table, const float *__restrict__ fc_out_w,
    float fc_out_b, float *__restrict__ logits, unsigned *status) {
  constexpr bool X3 = TERMS == 3;
  __shared__ __attribute__((aligned(16))) float s_tab[ROWS * H];  // 23 KB
  __shared__ __attribute__((aligned(16))) float s_wp[H * 3];
  __shared__ __attribute__((aligned(16))) float s_wo[H];

  const int tile = blockIdx.x;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int half = lane >> 5, n = lane & 31;
  const int prop = tile_prop[tile];
  if (prop < 0) return;  // padding tile (whole workgroup, before any barrier)

  {  // stage the per-proposal table + first/last layer weights
    const f32x4 *src = reinterpret_cast<const f32x4 *>(table + (size_t)prop * ROWS * H);
    f32x4 *dst = reinterpret_cast<f32x4 *>(s_tab);
    for (int i = t; i < ROWS * H / 4; i += 256) dst[i] = src[i];
    for (int i = t; i < H * 3; i += 256) s_wp[i] = fc_p_w[i];
    if (t < H) s_wo[t] = fc_out_w[t];
  }
  __syncthreads();

  const size_t pidx = (size_t)tile * TILE + wave * 32 + n;
  const size_t sidx = (size_t)(tile_src ? tile_src[tile] : tile) * TILE + wave * 32 + n;
  const float px = pts[sidx * 3 + 0], py = pts[sidx * 3 + 1], pz = pts[sidx * 3 + 2];

  // ---- fc_p (+ fc_z bias): H' = (Wp p + bp + zb) 2^KH, in accumulator layout
  f32x16 Hs[8];
#pragma unroll
  for (int ob = 0; ob < 8; ++ob) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ch = 32 * ob + 8 * (r >> 2) + 4 * half + (r & 3);
      float v = s_tab[ch];
      v = __builtin_fmaf(s_wp[ch * 3 + 0], px, v);
      v = __builtin_fmaf(s_wp[ch * 3 + 1], py, v);
      v = __builtin_fmaf(s_wp[ch * 3 + 2], pz, v);
      Hs[ob][r] = v;
    }
  }

  float amax = 0.f;
  half8 ahi[16], alo[16];  // B fragments of the block input, ks = 0..15
  for (int blk = 0; blk < NB; ++blk) {
    const float *S0 = s_tab + (1 + 4 * blk) * H, *T0 = S0 + H, *S1 = T0 + H, *T1 = S1 + H;
    // a' = relu(S0' H' + T0') for all 256 channels
#pragma unroll
    for (int kb = 0; kb < 8; ++kb)
      cbn_relu_split<X3>(Hs[kb], S0, T0, 32 * kb + 4 * half, ahi[2 * kb], ahi[2 * kb + 1],
                         alo[2 * kb], alo[2 * kb + 1], amax);
    for (int mb = 0; mb < 8; ++mb) {
      const half8 *w = packed + ((size_t)(blk * 8 + mb) * FRAGS_PER_CHUNK) * 64 + lane;
      // ---- GEMM1: 32 output channels of fc_0 over K = 256
      f32x16 acc = {0.f};
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        const half8 whi = w[(size_t)(2 * ks) * 64];
        acc = mfma(whi, ahi[ks], acc);
        if (X3) {
          const half8 wlo = w[(size_t)(2 * ks + 1) * 64];
          acc = mfma(whi, alo[ks], acc);
          acc = mfma(wlo, ahi[ks], acc);
        }
      }
      // a2' = relu(S1' acc + T1') for these 32 channels = 2 k-steps of GEMM2
      half8 bhi0, bhi1, blo0, blo1;
      cbn_relu_split<X3>(acc, S1, T1, 32 * mb + 4 * half, bhi0, bhi1, blo0, blo1, amax);
      // ---- GEMM2 partial: H'[ob] += fc_1[32ob.., 32mb..32mb+31] a2'
      const half8 *w2 = w + (size_t)32 * 64;
#pragma unroll
      for (int ob = 0; ob < 8; ++ob) {
        {
          const half8 whi = w2[(size_t)(4 * ob + 0) * 64];
          Hs[ob] = mfma(whi, bhi0, Hs[ob]);
          if (X3) {
            const half8 wlo = w2[(size_t)(4 * ob + 1) * 64];
            Hs[ob] = mfma(whi, blo0, Hs[ob]);
            Hs[ob] = mfma(wlo, bhi0, Hs[ob]);
          }
        }
        {
          const half8 whi = w2[(size_t)(4 * ob + 2) * 64];
          Hs[ob] = mfma(whi, bhi1, Hs[ob]);
          if (X3) {
            const half8 wlo = w2[(size_t)(4 * ob + 3) * 64];
            Hs[ob] = mfma(whi, blo1, Hs[ob]);
            Hs[ob] = mfma(wlo, bhi1, Hs[ob]);
          }
        }
      }
    }
  }

  // ---- out = fc_out(relu(CBN_f(h)))  (occ_decoder.py:120)
  const float *Sf = s_tab + 21 * H, *Tf = Sf + H;
  float part = 0.f;
#pragma unroll
  for (int ob = 0; ob < 8; ++ob) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ch0 = 32 * ob + 8 * q + 4 * half;
      const f32x4 s4 = *reinterpret_cast<const f32x4 *>(Sf + ch0);
      const f32x4 t4 = *reinterpret_cast<const f32x4 *>(Tf + ch0);
      const f32x4 w4 = *reinterpret_cast<const f32x4 *>(s_wo + ch0);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float a = __builtin_fmaf(s4[e], Hs[ob][4 * q + e], t4[e]);
        a = a > 0.f ? a : 0.f;
        part = __builtin_fmaf(w4[e], a, part);
      }
    }
  }
  part += __shfl_xor(part, 32);
  if (half == 0) logits[pidx] = part + fc_out_b;
  if (amax * 1.0f > 60000.f) atomicOr(status, 2u);  // f16 range exceeded
}

}  // namespace

RFD_API size_t rfd_occ_packed_bytes(void) { return PACKED_HALVES * sizeof(_Float16); }

RFD_API int rfd_occ_pack_weights(const float *fc0_w, const float *fc1_w,
                                 const int *kw0, int kw1, void *packed,
                                 void *stream) {
  const int threads = 256;
  const int blocks = (int)((PACKED_HALVES + threads - 1) / threads);
  hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(threads), 0,
                     (hipStream_t)stream, fc0_w, fc1_w, kw0[0], kw0[1], kw0[2],
                     kw0[3], kw0[4], kw1, (_Float16 *)packed);
  RFD_CHECK_LAUNCH();
  return 0;
}

RFD_API int rfd_occ_decode(int n_tiles, const float *pts, const int *tile_prop,
                           const int *tile_src, const void *packed, const float *fc_p_w,
                           const float *table, const float *fc_out_w,
                           float fc_out_b, float *logits, int mode,
                           void *stream) {
  if (n_tiles <= 0) return 0;
  RfdWorkspace *ws;
  int rc = rfd_get_workspace(&ws);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  if (mode == RFD_OCC_MODE_F16X3) {
    hipLaunchKernelGGL(occ_decode_kernel<3>, dim3(n_tiles), dim3(256), 0, s, n_tiles, pts,
                       tile_prop, tile_src, (const half8 *)packed, fc_p_w, table, fc_out_w, fc_out_b,
                       logits, ws->status);
  } else if (mode == RFD_OCC_MODE_F16X1) {
    hipLaunchKernelGGL(occ_decode_kernel<1>, dim3(n_tiles), dim3(256), 0, s, n_tiles, pts,
                       tile_prop, tile_src, (const half8 *)packed, fc_p_w, table, fc_out_w, fc_out_b,
                       logits, ws->status);
  } else {
    rfd_set_error("rfd_occ_decode: unknown mode", hipErrorInvalidValue);
    return (int)hipErrorInvalidValue;
  }
  RFD_CHECK_LAUNCH();
  return 0;
}
