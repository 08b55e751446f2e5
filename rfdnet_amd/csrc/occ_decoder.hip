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

// ---- activation epilogue, in slices -------------------------------------------
// relu(s*x + t) of the 16 accumulator values of one 32-channel block, converted
// to the next GEMM's B fragments.  Channel of register r: 8(r>>2) + 4*half +
// (r&3), so the table rows are read as four float4.  The work is cut into 8
// two-value SLICES so it can be issued in the shadow of another GEMM's MFMAs.
struct EpiTab {
  f32x4 s[4], t[4];
};

__device__ __forceinline__ EpiTab load_epi_tab(const float *s_row, const float *t_row, int ch0) {
  EpiTab e;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    e.s[q] = *reinterpret_cast<const f32x4 *>(s_row + ch0 + 8 * q);
    e.t[q] = *reinterpret_cast<const f32x4 *>(t_row + ch0 + 8 * q);
  }
  return e;
}

// running max of packed non-negative f16 pairs, compared as u16 (monotone)
__device__ __forceinline__ unsigned pk_max_u16(unsigned a, unsigned b) {
  unsigned r;
  asm("v_pk_max_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

template <bool WITH_LO>
__device__ __forceinline__ void epi_slice(int i, const f32x16 &x, const EpiTab &tb, unsigned (&hiw)[8],
                                          unsigned (&low)[8], unsigned &amax16) {
  const int q = i >> 1, e0 = 2 * (i & 1);
  float a0 = __builtin_fmaf(tb.s[q][e0], x[2 * i], tb.t[q][e0]);
  float a1 = __builtin_fmaf(tb.s[q][e0 + 1], x[2 * i + 1], tb.t[q][e0 + 1]);
  a0 = a0 > 0.f ? a0 : 0.f;
  a1 = a1 > 0.f ? a1 : 0.f;
  const half2v h2 = __builtin_bit_cast(half2v, __builtin_amdgcn_cvt_pkrtz(a0, a1));  // round to zero
  hiw[i] = __builtin_bit_cast(unsigned, h2);
  amax16 = pk_max_u16(amax16, hiw[i]);
  if (WITH_LO) {
    // a - (float)hi (exact): one v_fma_mix_f32 per value, reading the f16 halves in place
    float r0, r1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hiw[i]), "v"(a0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hiw[i]), "v"(a1));
    low[i] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(r0, r1));
  } else {
    low[i] = 0u;
  }
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ half8 words_to_frag(const unsigned (&w)[8], int o) {
  u32x4 v = {w[o], w[o + 1], w[o + 2], w[o + 3]};
  return __builtin_bit_cast(half8, v);
}

// LDS carve-up (ONE static object: a second __shared__ object makes hipcc drain
// vmcnt before every ds_read of an LDS-DMA pipeline; the base stays 16-B
// aligned):  [ table 23x256 f32 | fc_p 256x3 f32 | fc_out 256 f32 | ring of
// four 32-KiB half-chunks ]  = 27 648 + 131 072 = 158 720 B of the CU's 160 KiB.
constexpr int SMEM_TAB_BYTES = (ROWS * H + H * 3 + H) * 4;
constexpr int HALF_FRAGS = 32;                                // fragments per half-chunk
constexpr int HALF_BYTES = HALF_FRAGS * FRAG_HALVES * 2;      // 32 KiB
constexpr int N_HALVES = NB * 8 * 2;                          // 80
constexpr int SMEM_BYTES = SMEM_TAB_BYTES + 4 * HALF_BYTES;

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

// One 1-KiB LDS-DMA piece (global_load_lds_dwordx4: lane-linear destination =
// the fragment layout).  Half-chunk h = fragments [32h, 32h+32) of the stream:
// even h = fc_0 fragments of chunk h/2, odd h = fc_1 fragments.  Wave w moves
// fragments 8w..8w+7 of a half.
__device__ __forceinline__ void dma_piece(const half8 *__restrict__ packed, unsigned char *s_slots,
                                          int h, int j, int wave, int lane) {
  const int frag = wave * 8 + j;
  const int hs = h >= N_HALVES ? h - N_HALVES : h;  // halves >= 80 belong to the NEXT tile (same weights)
  __builtin_amdgcn_global_load_lds(
      (gbl_void *)(packed + ((size_t)hs * HALF_FRAGS + frag) * 64 + lane),
      (lds_void *)(s_slots + (h & 3) * HALF_BYTES + frag * 1024), 16, 0, 0);
}

template <int TERMS>
__global__ __launch_bounds__(256) void occ_decode_kernel(
    int n_tiles, const float *__restrict__ pts, const int *__restrict__ tile_prop,
    const int *__restrict__ tile_src, const half8 *__restrict__ packed, const float *__restrict__ fc_p_w,
    const float *__restrict__ table, const float *__restrict__ fc_out_w,
    float fc_out_b, float *__restrict__ logits, unsigned *status, int tiles_per_wg) {
  constexpr bool X3 = TERMS == 3;
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM_BYTES];
  float *s_tab = reinterpret_cast<float *>(smem);
  float *s_wp = s_tab + ROWS * H;
  float *s_wo = s_wp + H * 3;
  unsigned char *s_slots = smem + SMEM_TAB_BYTES;

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int half = lane >> 5, n = lane & 31;
  unsigned amax16 = 0u;

  // Persistent workgroup: a contiguous run of tiles (mostly one proposal, so the
  // table is staged rarely) and ONE uninterrupted weight ring -- the last two
  // iterations of a tile already fetch the first three half-chunks of the next.
  const int t_begin = blockIdx.x * tiles_per_wg;
  const int t_end = (t_begin + tiles_per_wg) < n_tiles ? (t_begin + tiles_per_wg) : n_tiles;
  int cur_prop = -1;
  bool ring_primed = false;
  for (int tile = t_begin; tile < t_end; ++tile) {
  const int prop = tile_prop[tile];
  if (prop < 0) continue;  // padding tile (wave-uniform)
  const bool has_next = tile + 1 < t_end;

  const size_t pidx = (size_t)tile * TILE + wave * 32 + n;
  const size_t sidx = (size_t)(tile_src ? tile_src[tile] : tile) * TILE + wave * 32 + n;
  const float px = pts[sidx * 3 + 0], py = pts[sidx * 3 + 1], pz = pts[sidx * 3 + 2];

  if (!ring_primed) {  // weights of chunk 0 (fc_0, fc_1) and chunk 1 (fc_0) in flight first
#pragma unroll
    for (int j = 0; j < 24; ++j) dma_piece(packed, s_slots, j >> 3, j & 7, wave, lane);
  }
  if (prop != cur_prop) {  // stage the per-proposal table (+ first/last layer weights once)
    __syncthreads();       // everyone is done with the previous proposal's table
    const f32x4 *src = reinterpret_cast<const f32x4 *>(table + (size_t)prop * ROWS * H);
    f32x4 *dst = reinterpret_cast<f32x4 *>(s_tab);
    for (int i = t; i < ROWS * H / 4; i += 256) dst[i] = src[i];
    if (!ring_primed) {
      for (int i = t; i < H * 3; i += 256) s_wp[i] = fc_p_w[i];
      if (t < H) s_wo[t] = fc_out_w[t];
    }
    cur_prop = prop;
  }
  ring_primed = true;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

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

  half8 ahi[16], alo[16];  // B fragments of the block input, ks = 0..15
  for (int blk = 0; blk < NB; ++blk) {
    const float *S0 = s_tab + (1 + 4 * blk) * H, *T0 = S0 + H, *S1 = T0 + H, *T1 = S1 + H;
    // ---- a' = relu(S0' H' + T0') for all 256 channels, fused with fc_0 of the
    // block's first 32 output channels: k-steps 2kb, 2kb+1 only need channel
    // block kb, so block kb+1 is converted in the shadow of their six MFMAs.
    f32x16 acc_cur = {0.f};
    {
      const half8 *w = reinterpret_cast<const half8 *>(s_slots + ((2 * blk * 8) & 3) * HALF_BYTES) + lane;
      half8 fh = w[0], fl = w[X3 ? 64 : 0];
      {
        const EpiTab tb = load_epi_tab(S0, T0, 4 * half);
        unsigned hw[8], lw[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) epi_slice<X3>(i, Hs[0], tb, hw, lw, amax16);
        ahi[0] = words_to_frag(hw, 0); ahi[1] = words_to_frag(hw, 4);
        alo[0] = words_to_frag(lw, 0); alo[1] = words_to_frag(lw, 4);
      }
#pragma unroll
      for (int kb = 0; kb < 8; ++kb) {
        unsigned hw[8], lw[8];
        EpiTab tb;
        if (kb < 7) tb = load_epi_tab(S0, T0, 32 * (kb + 1) + 4 * half);
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
          const int ks = 2 * kb + sub;
          const half8 ch = fh, cl = fl;
          if (ks < 15) {
            fh = w[(2 * ks + 2) * 64];
            if (X3) fl = w[(2 * ks + 3) * 64];
          }
          acc_cur = mfma(ch, ahi[ks], acc_cur);
          if (X3) {
            acc_cur = mfma(ch, alo[ks], acc_cur);
            acc_cur = mfma(cl, ahi[ks], acc_cur);
          }
          if (kb < 7) {
#pragma unroll
            for (int i = 4 * sub; i < 4 * sub + 4; ++i) epi_slice<X3>(i, Hs[kb + 1], tb, hw, lw, amax16);
          }
          // issue order inside the step: each MFMA occupies the matrix pipe for
          // 32 cycles and the next one (same accumulator) cannot start earlier,
          // so independent LDS reads / VALU are slotted into those gaps
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
          if (X3) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if (kb < 7) {
          ahi[2 * kb + 2] = words_to_frag(hw, 0); ahi[2 * kb + 3] = words_to_frag(hw, 4);
          alo[2 * kb + 2] = words_to_frag(lw, 0); alo[2 * kb + 3] = words_to_frag(lw, 4);
        }
      }
    }
    // the slot just read is refilled by this iteration's LDS-DMA below
    __syncthreads();

    for (int mb = 0; mb < 8; ++mb) {
      const int c = blk * 8 + mb;  // global chunk
      // ---- phase A: epilogue of fc_0 block mb, in the shadow of fc_0 block mb+1
      const EpiTab tb = load_epi_tab(S1, T1, 32 * mb + 4 * half);
      unsigned hw[8], lw[8];
      f32x16 acc_next = {0.f};
      half8 g[8];  // fc_1 fragments of the current pair of output blocks
      const half8 *w2 = reinterpret_cast<const half8 *>(s_slots + ((2 * c + 1) & 3) * HALF_BYTES) + lane;
      if (mb < 7) {
        const half8 *w = reinterpret_cast<const half8 *>(s_slots + ((2 * c + 2) & 3) * HALF_BYTES) + lane;
        half8 fh = w[0], fl = w[X3 ? 64 : 0];
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
          const half8 ch = fh, cl = fl;
          if (ks < 15) {
            fh = w[(2 * ks + 2) * 64];
            if (X3) fl = w[(2 * ks + 3) * 64];
          } else {  // last step: start fetching fc_1's first fragments
#pragma unroll
            for (int q = 0; q < 8; ++q)
              if (X3 || !(q & 1)) g[q] = w2[q * 64];
          }
          acc_next = mfma(ch, ahi[ks], acc_next);
          if (X3) {
            acc_next = mfma(ch, alo[ks], acc_next);
            acc_next = mfma(cl, ahi[ks], acc_next);
          }
          if (ks < 8) epi_slice<X3>(ks, acc_cur, tb, hw, lw, amax16);  // VALU under the MFMAs
          {  // ONE LDS-DMA piece per step: four waves x 1 KiB per 96-cycle step keeps the
             // address path at ~2/3 load (two per step in half of the steps saturated it and
             // every transfer then cost its wave ~50 cycles of MFMA issue)
            const int h = 2 * c + 3 + (ks >> 3);
            if (h < N_HALVES || has_next) dma_piece(packed, s_slots, h, ks & 7, wave, lane);
          }
          // M r r v.. | M v.. | M v.. D : everything that is independent of the
          // accumulator chain goes into the 32-cycle gaps behind each MFMA
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (ks < 15) __builtin_amdgcn_sched_group_barrier(0x100, X3 ? 2 : 1, 0);
          else __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
          if (ks < 8) {
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
            if (X3) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
            }
          } else {
            if (X3) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            }
          }
          __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
        for (int q = 0; q < 8; ++q)
          if (X3 || !(q & 1)) g[q] = w2[q * 64];
#pragma unroll
        for (int i = 0; i < 8; ++i) epi_slice<X3>(i, acc_cur, tb, hw, lw, amax16);
#pragma unroll
        for (int j = 0; j < 16; ++j) {  // next block's first halves
          const int h = 2 * c + 3 + (j >> 3);
          if (h < N_HALVES || has_next) dma_piece(packed, s_slots, h, j & 7, wave, lane);
        }
      }
      const half8 bhi0 = words_to_frag(hw, 0), bhi1 = words_to_frag(hw, 4);
      const half8 blo0 = words_to_frag(lw, 0), blo1 = words_to_frag(lw, 4);
      // ---- phase B: H'[ob] += fc_1[32ob.., 32mb..32mb+31] a2', two output blocks
      // interleaved; the next pair's fragments are fetched under the 12 MFMAs.
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        half8 cf[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) cf[q] = g[q];
        if (p < 3) {
#pragma unroll
          for (int q = 0; q < 8; ++q)
            if (X3 || !(q & 1)) g[q] = w2[(8 * (p + 1) + q) * 64];
        }
        const int o0 = 2 * p, o1 = 2 * p + 1;
        Hs[o0] = mfma(cf[0], bhi0, Hs[o0]);
        Hs[o1] = mfma(cf[4], bhi0, Hs[o1]);
        if (X3) {
          Hs[o0] = mfma(cf[0], blo0, Hs[o0]);
          Hs[o1] = mfma(cf[4], blo0, Hs[o1]);
          Hs[o0] = mfma(cf[1], bhi0, Hs[o0]);
          Hs[o1] = mfma(cf[5], bhi0, Hs[o1]);
        }
        Hs[o0] = mfma(cf[2], bhi1, Hs[o0]);
        Hs[o1] = mfma(cf[6], bhi1, Hs[o1]);
        if (X3) {
          Hs[o0] = mfma(cf[2], blo1, Hs[o0]);
          Hs[o1] = mfma(cf[6], blo1, Hs[o1]);
          Hs[o0] = mfma(cf[3], bhi1, Hs[o0]);
          Hs[o1] = mfma(cf[7], bhi1, Hs[o1]);
        }
        if (p < 3) {
#pragma unroll
          for (int q = 0; q < (X3 ? 8 : 4); ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // prefetched halves landed + everyone done with the slots refilled next
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      acc_cur = acc_next;
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
  }  // persistent tile loop
  // 0x7bff = 65504 = largest finite f16: the round-to-zero conversion saturates there
  if ((amax16 & 0xffffu) >= 0x7bffu || (amax16 >> 16) >= 0x7bffu) atomicOr(status, 2u);
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
  // one persistent workgroup per CU (the kernel owns the CU: 158 KiB LDS, 1 wave/SIMD)
  const int ncu = ws->num_cu > 0 ? ws->num_cu : 256;
  const int tiles_per_wg = ceil_div(n_tiles, ncu);
  const int grid = ceil_div(n_tiles, tiles_per_wg);
  if (mode == RFD_OCC_MODE_F16X3) {
    hipLaunchKernelGGL(occ_decode_kernel<3>, dim3(grid), dim3(256), 0, s, n_tiles, pts,
                       tile_prop, tile_src, (const half8 *)packed, fc_p_w, table, fc_out_w, fc_out_b,
                       logits, rfd_status_word(ws, s), tiles_per_wg);
  } else if (mode == RFD_OCC_MODE_F16X1) {
    hipLaunchKernelGGL(occ_decode_kernel<1>, dim3(grid), dim3(256), 0, s, n_tiles, pts,
                       tile_prop, tile_src, (const half8 *)packed, fc_p_w, table, fc_out_w, fc_out_b,
                       logits, rfd_status_word(ws, s), tiles_per_wg);
  } else {
    rfd_set_error("rfd_occ_decode: unknown mode", hipErrorInvalidValue);
    return (int)hipErrorInvalidValue;
  }
  RFD_CHECK_LAUNCH();
  return 0;
}
