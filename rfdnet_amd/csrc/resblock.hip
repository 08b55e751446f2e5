// resblock.hip -- one whole ResnetBlockFC of the skip-propagation point encoder
// as a single MFMA kernel.
//
// Replaces ResnetBlockFC.forward (models/iscnet/modules/layers.py:39-48) as
// called per point by ResnetPointnet.forward (layers.py:364-392), with the
// pooled half of the block input already reduced to one vector per proposal by
// the host (the block input is cat([net, pooled]); the pooled half is constant
// over a proposal's points):
//
//   a   = relu(x)                         x: (M, K_IN) rows, K_IN = 256 | 512
//   h   = W0 a + g0[group]                g0 = W0_pooled relu(pooled) + b0
//   out = Ws a + W1 relu(h) + gs[group]   gs = Ws_pooled relu(pooled) + b1
//
// (the reference's in-place ReLU makes the shortcut see relu(x) too.)
//
// Decomposition.  A workgroup (4 waves) takes 128 rows; each WAVE owns 32 rows x
// all output channels.  v_mfma_f32_32x32x16_f16 computes D[channel, row]; the
// sixteen 32-channel output blocks of [W0 ; Ws] are sixteen INDEPENDENT
// accumulators (256 accumulator registers per lane), so the first GEMM runs
// K-outer: per 16-wide k-step the lane converts 8 of its row's inputs to one B
// fragment pair (hi, lo) and issues 48 back-to-back MFMAs.  relu(h) never leaves
// the lane: the accumulator layout of block mb is the B-fragment layout of fc_1's
// k-slab mb (the pack kernel permutes fc_1's columns accordingly, as in the
// occupancy decoder), and fc_1 accumulates straight onto the shortcut blocks.
//
// Weights stream through a 4-slot ring of 32-KiB pieces in LDS (LDS-DMA, three
// pieces ahead); piece = one k-step of [W0;Ws] (16 blocks x hi/lo) or one k-slab
// of W1 (8 blocks x 2 sub-steps x hi/lo).  x rows are prefetched four k-steps
// ahead into registers, across tile boundaries (persistent workgroups).
//
// Precision: both operands split into f16 (hi, lo), hi*hi + hi*lo + lo*hi with
// fp32 accumulate, operands pre-scaled by powers of two -- same scheme as
// occ_decoder.hip / gemm_f16x3.hip.
//
// Roofline: 3 * 2 * 256 * 256 FLOP per row (K_IN = 256) against 2 KiB of HBM
// traffic per row => MFMA bound (3x issued work), HBM ~40 % busy at that rate.
#include "common.h"
#include "../../include/rfd_occ.h"
#include <utility>

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

constexpr int HID = RFD_RESBLOCK_HIDDEN;      // 256 output channels / hidden width
constexpr int TILE = RFD_RESBLOCK_TILE;       // 128 rows per workgroup tile
constexpr int KA = RFD_RESBLOCK_KA;           // activations scaled by 2^KA before the split
constexpr int PIECE_FRAGS = 32;               // 1-KiB fragments per ring piece
constexpr int PIECE_BYTES = PIECE_FRAGS * 1024;
constexpr int XD = 4;                         // x prefetch distance (k-steps)
constexpr int XS = XD + 1;                    // x register slots
constexpr int SMEM_BYTES = 2 * HID * 4 + 4 * PIECE_BYTES;

// ---- weight packing: stream order = consumption order -------------------------------
// piece ks < KS:   fragment f = 2*ob + s, ob < 8: fc_0 rows 32ob.., ob >= 8: shortcut rows
//                  32(ob-8)..; A[m][k] with k = 16ks + 8h + j, lane = 32h + m
// piece KS + mb:   fragment f = 4*ob + 2*sub + s: fc_1 rows 32ob.., columns
//                  32mb + 16sub + 8(j>>2) + 4h + (j&3)   (accumulator order of block mb)
// s = 0: f16 hi (round to nearest) of w * 2^kw, s = 1: lo = remainder.
__global__ void resblock_pack_kernel(const float *__restrict__ w0, const float *__restrict__ ws,
                                     const float *__restrict__ w1, int k_in, int ld01, int kw0, int kw1,
                                     _Float16 *__restrict__ packed, size_t total) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int j = e & 7, lane = (e >> 3) & 63;
  const int frag = (int)(e >> 9);
  const int f = frag & 31, piece = frag >> 5;
  const int m = lane & 31, h = lane >> 5;
  const int KS = k_in / 16;
  float w;
  int kw, s;
  if (piece < KS) {
    const int ob = f >> 1;
    s = f & 1;
    const int out_ch = 32 * (ob & 7) + m, in_ch = 16 * piece + 8 * h + j;
    w = (ob < 8 ? w0 : ws)[(size_t)out_ch * ld01 + in_ch];
    kw = ob < 8 ? kw0 : kw1;
  } else {
    const int mb = piece - KS, ob = f >> 2, sub = (f >> 1) & 1;
    s = f & 1;
    const int out_ch = 32 * ob + m, in_ch = 32 * mb + 16 * sub + 8 * (j >> 2) + 4 * h + (j & 3);
    w = w1[(size_t)out_ch * HID + in_ch];
    kw = kw1;
  }
  w = ldexpf(w, kw);
  const _Float16 hi = (_Float16)w;
  const _Float16 lo = (_Float16)(w - (float)hi);
  packed[e] = s == 0 ? hi : lo;
}

__device__ __forceinline__ f32x16 mfma(half8 a, half8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ unsigned pk_max_u16(unsigned a, unsigned b) {
  unsigned r;
  asm("v_pk_max_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// two non-negative fp32 values -> packed f16 hi (round to zero) and lo words
__device__ __forceinline__ void split2(float a0, float a1, unsigned &hiw, unsigned &low, unsigned &amax16) {
  const half2v h2 = __builtin_bit_cast(half2v, __builtin_amdgcn_cvt_pkrtz(a0, a1));
  hiw = __builtin_bit_cast(unsigned, h2);
  amax16 = pk_max_u16(amax16, hiw);
  const float r0 = __builtin_fmaf((float)h2[0], -1.0f, a0), r1 = __builtin_fmaf((float)h2[1], -1.0f, a1);
  low = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(r0, r1));
}

__device__ __forceinline__ half8 words_to_frag(unsigned a, unsigned b, unsigned c, unsigned d) {
  u32x4 v = {a, b, c, d};
  return __builtin_bit_cast(half8, v);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// B fragments (2 sub-steps x hi/lo) of one fc_1 k-slab = relu(h) of one fc_0 block
struct A2 {
  half8 hi0, hi1, lo0, lo1;
};

// does step s (tile-local, wraps) issue x loads?
template <int K_IN>
__device__ __forceinline__ constexpr int has_xload(int s) {
  constexpr int KS = K_IN / 16, NP = KS + 8;
  s = (s + NP) % NP;
  return (s + XD < KS || s >= NP - XD) ? 1 : 0;
}

// Per-wave state of one tile; step<I>() is instantiated for every ring piece so that
// every register-array index below is a compile-time constant.
template <int K_IN>
struct Tile {
  static constexpr int KS = K_IN / 16;   // k-steps of the first GEMM
  static constexpr int NP = KS + 8;      // ring pieces per tile
  static_assert(NP % 4 == 0, "ring slot = piece & 3 across tiles");

  f32x16 acc[16];
  f32x4 xr[XS][2];
  A2 a2;
  half8 bhi, blo;          // B fragments of the current k-step (converted one step ahead)
  unsigned nhw[4], nlw[4]; // ... of the next k-step, being converted
  unsigned ehw[8], elw[8]; // relu(h) words of the next fc_1 k-slab, being converted
  unsigned amax16;
  const half8 *packed;
  unsigned char *s_ring;
  const float *s_g0, *s_gs;
  const float *xp, *xn;    // this lane's row in the current / next tile (+ 8*half)
  float *op;               // this lane's output row (+ 4*half)
  float inv_s1, inv_out;
  int wave, lane, half;
  unsigned lane16;         // lane * 16: the per-lane part of every LDS-DMA source address
  bool first;

  // wave-uniform source address (scalar base, re-derived every tile so that the 8 x NP
  // constant offsets are not hoisted into -- and spilled from -- vector registers)
  __device__ __forceinline__ void dma1(int piece, int slot, int jj) {
    const char *src = reinterpret_cast<const char *>(packed) + ((size_t)piece * PIECE_FRAGS + wave * 8) * 1024;
    __builtin_amdgcn_global_load_lds((gbl_void *)(src + jj * 1024 + lane16),
                                     (lds_void *)(s_ring + slot * PIECE_BYTES + (wave * 8 + jj) * 1024), 16, 0, 0);
  }
  __device__ __forceinline__ void dma(int piece, int slot) {
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) dma1(piece, slot, jj);
  }

  // 8 inputs of one k-step.  Issued as opaque instructions: the compiler's own wait
  // insertion would otherwise drain vmcnt (and with it the LDS-DMA pipeline) before
  // the first use; step<I>()'s explicit wait covers these loads (issued BEFORE the
  // ring piece that the wait two steps later is for).
  template <int SL, int OFF_BYTES>
  __device__ __forceinline__ void load_x(const float *p) {
    asm volatile("global_load_dwordx4 %0, %2, off offset:%3\n\t"
                 "global_load_dwordx4 %1, %2, off offset:%4"
                 : "=&v"(xr[SL][0]), "=&v"(xr[SL][1])
                 : "v"(p), "n"(OFF_BYTES), "n"(OFF_BYTES + 16)
                 : "memory");
  }

  // relu(x) 2^KA of k-step slot `sl` -> (bhi, blo)
  __device__ __forceinline__ void convert_x(int sl) {
    const f32x4 x0 = xr[sl][0], x1 = xr[sl][1];
    unsigned hw[4], lw[4];
    const float sc = (float)(1 << KA);
    auto rs = [sc](float v) { return (v > 0.f ? v : 0.f) * sc; };
    split2(rs(x0[0]), rs(x0[1]), hw[0], lw[0], amax16);
    split2(rs(x0[2]), rs(x0[3]), hw[1], lw[1], amax16);
    split2(rs(x1[0]), rs(x1[1]), hw[2], lw[2], amax16);
    split2(rs(x1[2]), rs(x1[3]), hw[3], lw[3], amax16);
    bhi = words_to_frag(hw[0], hw[1], hw[2], hw[3]);
    blo = words_to_frag(lw[0], lw[1], lw[2], lw[3]);
  }

  // two of the 8 inputs of k-step slot SL -> word `i` of the next B pair
  __device__ __forceinline__ void conv_slice(int sl, int i) {
    const float sc = (float)(1 << KA);
    float v0 = xr[sl][i >> 1][2 * (i & 1)], v1 = xr[sl][i >> 1][2 * (i & 1) + 1];
    v0 = (v0 > 0.f ? v0 : 0.f) * sc;
    v1 = (v1 > 0.f ? v1 : 0.f) * sc;
    split2(v0, v1, nhw[i], nlw[i], amax16);
  }
  // two of the 16 values of an fc_0 accumulator block -> word `i` of the next a2
  __device__ __forceinline__ void epi_slice(const f32x16 &x, int ch_base, int i) {
    const int q = i >> 1, e0 = 2 * (i & 1);
    const float *g = s_g0 + ch_base + 8 * q + e0;
    float v0 = __builtin_fmaf(x[2 * i], inv_s1, g[0]), v1 = __builtin_fmaf(x[2 * i + 1], inv_s1, g[1]);
    v0 = v0 > 0.f ? v0 : 0.f;
    v1 = v1 > 0.f ? v1 : 0.f;
    split2(v0, v1, ehw[i], elw[i], amax16);
  }
  __device__ __forceinline__ void epi_finish() {
    a2.hi0 = words_to_frag(ehw[0], ehw[1], ehw[2], ehw[3]);
    a2.hi1 = words_to_frag(ehw[4], ehw[5], ehw[6], ehw[7]);
    a2.lo0 = words_to_frag(elw[0], elw[1], elw[2], elw[3]);
    a2.lo1 = words_to_frag(elw[4], elw[5], elw[6], elw[7]);
  }

  template <int I>
  __device__ __forceinline__ void step() {
    const half8 *w = reinterpret_cast<const half8 *>(s_ring + (I & 3) * PIECE_BYTES) + lane;
    // Fragments of the first group of four output blocks: their LDS latency hides behind
    // the issue below.  MFMA order everywhere: FOUR independent accumulators in rotation
    // (a dependent MFMA that is not issued back to back with its producer waits ~45 extra
    // cycles; with three or more other MFMAs in between it never waits).
    half8 cur[4][2], nxt[4][2];
    if (I < KS) {
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        cur[o][0] = w[(2 * o) * 64];
        cur[o][1] = w[(2 * o + 1) * 64];
      }
    } else {
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        cur[o][0] = w[(4 * o) * 64];
        cur[o][1] = w[(4 * o + 1) * 64];
      }
    }
    // ---- issue: x four k-steps ahead (the ring piece three ahead goes out one 1-KiB
    // transfer at a time between the MFMAs below)
    if (I + XD < KS) {
      load_x<(I + XD) % XS, 64 * ((I + XD) % KS)>(xp);
    } else if (I >= NP - XD) {
      constexpr int k = (I + XD) % NP;
      load_x<k % XS, 64 * (k % KS)>(xn);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (I < KS) {
      // ---- first GEMM, k-step I: [h ; shortcut] += [W0 ; Ws][:, 16I..16I+15] relu(x),
      // four output blocks at a time; the next group's fragments and the next k-step's
      // B pair are produced in the shadow of the 12 MFMAs
      const half8 bh = bhi, bl = blo;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (g < 3) {
#pragma unroll
          for (int o = 0; o < 4; ++o) {
            nxt[o][0] = w[(2 * (4 * g + 4 + o)) * 64];
            nxt[o][1] = w[(2 * (4 * g + 4 + o) + 1) * 64];
          }
        }
#pragma unroll
        for (int o = 0; o < 4; ++o) acc[4 * g + o] = mfma(cur[o][0], bh, acc[4 * g + o]);
#pragma unroll
        for (int o = 0; o < 4; ++o) acc[4 * g + o] = mfma(cur[o][0], bl, acc[4 * g + o]);
#pragma unroll
        for (int o = 0; o < 4; ++o) acc[4 * g + o] = mfma(cur[o][1], bh, acc[4 * g + o]);
        if (g == 0 && I + 1 < KS) {
#pragma unroll
          for (int i = 0; i < 4; ++i) conv_slice((I + 1) % XS, i);
        }
        if ((g == 1 || g == 2) && I == KS - 1) {
#pragma unroll
          for (int i = 0; i < 4; ++i) epi_slice(acc[0], 4 * half, 4 * (g - 1) + i);
        }
        dma1((I + 3) % NP, (I + 3) & 3, 2 * g);
        dma1((I + 3) % NP, (I + 3) & 3, 2 * g + 1);
#pragma unroll
        for (int q = 0; q < 12; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (q < 8) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          if (q == 5 || q == 10) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (g < 3) {
#pragma unroll
          for (int o = 0; o < 4; ++o) {
            cur[o][0] = nxt[o][0];
            cur[o][1] = nxt[o][1];
          }
        }
      }
      if (I + 1 < KS) {
        bhi = words_to_frag(nhw[0], nhw[1], nhw[2], nhw[3]);
        blo = words_to_frag(nlw[0], nlw[1], nlw[2], nlw[3]);
      }
      if (I == KS - 1) epi_finish();
    } else {
      // ---- second GEMM, k-slab mb: out += W1[:, 32mb..32mb+31] relu(h[mb]), four output
      // blocks at a time in two halves (sub-step 0: fragments 4ob, 4ob+1; sub-step 1:
      // 4ob+2, 4ob+3); the next slab's relu(h) is converted in the shadow of the MFMAs
      constexpr int mb = I - KS;
      const A2 b = a2;
#pragma unroll
      for (int hg = 0; hg < 4; ++hg) {   // (group of 4 blocks, sub-step) = (hg >> 1, hg & 1)
        const int g = hg >> 1, sub = hg & 1;
        if (hg < 3) {
          const int ng = (hg + 1) >> 1, nsub = (hg + 1) & 1;
#pragma unroll
          for (int o = 0; o < 4; ++o) {
            nxt[o][0] = w[(4 * (4 * ng + o) + 2 * nsub) * 64];
            nxt[o][1] = w[(4 * (4 * ng + o) + 2 * nsub + 1) * 64];
          }
        }
        const half8 bhq = sub ? b.hi1 : b.hi0, blq = sub ? b.lo1 : b.lo0;
#pragma unroll
        for (int o = 0; o < 4; ++o) acc[8 + 4 * g + o] = mfma(cur[o][0], bhq, acc[8 + 4 * g + o]);
#pragma unroll
        for (int o = 0; o < 4; ++o) acc[8 + 4 * g + o] = mfma(cur[o][0], blq, acc[8 + 4 * g + o]);
#pragma unroll
        for (int o = 0; o < 4; ++o) acc[8 + 4 * g + o] = mfma(cur[o][1], bhq, acc[8 + 4 * g + o]);
        if (mb < 7) {
          epi_slice(acc[(mb + 1) & 7], 32 * (mb + 1) + 4 * half, 2 * hg);
          epi_slice(acc[(mb + 1) & 7], 32 * (mb + 1) + 4 * half, 2 * hg + 1);
        }
        dma1((I + 3) % NP, (I + 3) & 3, 2 * hg);
        dma1((I + 3) % NP, (I + 3) & 3, 2 * hg + 1);
#pragma unroll
        for (int q = 0; q < 12; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (q < 8) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          if (q == 5 || q == 10) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (hg < 3) {
#pragma unroll
          for (int o = 0; o < 4; ++o) {
            cur[o][0] = nxt[o][0];
            cur[o][1] = nxt[o][1];
          }
        }
      }
      if (mb < 7) epi_finish();
    }
    if (I == NP - 1) {
      // ---- out = acc 2^-(KA+kw1) + gs, the rows of this wave, 16-B stores
#pragma unroll
      for (int ob = 0; ob < 8; ++ob) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 g = *reinterpret_cast<const f32x4 *>(s_gs + 32 * ob + 8 * q + 4 * half);
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(acc[8 + ob][4 * q + e], inv_out, g[e]);
          *reinterpret_cast<f32x4 *>(op + 32 * ob + 8 * q) = v;
        }
      }
      // next tile's first B pair (its x(0) was loaded XD steps ago)
      convert_x(0);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- piece I+1 (and the x loads issued just before it) landed -- issued two steps
    // ago.  VMEM ops that may stay in flight: two newer DMA batches, the x loads of
    // steps I-1 and I, and the 32 row stores of this tile (last step) / of the
    // previous tile (first two steps)
    constexpr int nxl = 2 * (has_xload<K_IN>(I - 1) + has_xload<K_IN>(I));
    if (I == NP - 1) {
      wait_vmcnt<16 + nxl + 32>();
    } else if (I < 2) {
      if (first) wait_vmcnt<16 + nxl>();
      else wait_vmcnt<16 + nxl + 32>();
    } else {
      wait_vmcnt<16 + nxl>();
    }
    __builtin_amdgcn_s_barrier();
  }
};

template <int K_IN, int... I>
__device__ __forceinline__ void run_steps(Tile<K_IN> &tl, std::integer_sequence<int, I...>) {
  (tl.template step<I>(), ...);
}

template <int K_IN>
__global__ __launch_bounds__(256) void resblock_kernel(
    int n_tiles, int tiles_per_group, const float *__restrict__ x, const half8 *__restrict__ packed,
    const float *__restrict__ g0, const float *__restrict__ gs, float *__restrict__ out, float inv_s1,
    float inv_out, unsigned *status, int tiles_per_wg) {
  typedef Tile<K_IN> T;
  // ONE static LDS object (see occ_decoder.hip): [ g0*2^KA | gs | ring 4 x 32 KiB ]
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM_BYTES];
  float *s_g0 = reinterpret_cast<float *>(smem);
  float *s_gs = s_g0 + HID;

  const int t = threadIdx.x;
  T tl;
  tl.lane = t & 63;
  tl.wave = __builtin_amdgcn_readfirstlane(t >> 6);
  tl.half = tl.lane >> 5;
  tl.lane16 = (unsigned)tl.lane * 16u;
  const int n = tl.lane & 31;
  tl.amax16 = 0u;
  tl.packed = packed;
  tl.s_ring = smem + 2 * HID * 4;
  tl.s_g0 = s_g0;
  tl.s_gs = s_gs;
  tl.inv_s1 = inv_s1;
  tl.inv_out = inv_out;

  const int t_begin = blockIdx.x * tiles_per_wg;
  const int t_end = (t_begin + tiles_per_wg) < n_tiles ? (t_begin + tiles_per_wg) : n_tiles;
  if (t_begin >= t_end) return;
  auto xrow = [&](int tile) { return x + ((size_t)tile * TILE + tl.wave * 32 + n) * K_IN + 8 * tl.half; };

  // prologue, in the steady-state issue order: x(0); x(1), DMA 0; x(2), DMA 1; x(3), DMA 2
  {
    const float *xp = xrow(t_begin);
    tl.packed = packed;
    tl.template load_x<0, 0>(xp);
    tl.template load_x<1, 64>(xp);
    tl.dma(0, 0);
    tl.template load_x<2, 128>(xp);
    tl.dma(1, 1);
    tl.template load_x<3, 192>(xp);
    tl.dma(2, 2);
    wait_vmcnt<20>();      // piece 0 and x(0), x(1) landed
    __builtin_amdgcn_s_barrier();
    tl.convert_x(0);
  }

  int cur_grp = -1;
  for (int tile = t_begin; tile < t_end; ++tile) {
    tl.first = tile == t_begin;
    const int grp = tile / tiles_per_group;
    if (grp != cur_grp) {  // stage the per-group vectors (rare: once per proposal)
      __syncthreads();
      s_g0[t] = g0[(size_t)grp * HID + t] * (float)(1 << KA);
      s_gs[t] = gs[(size_t)grp * HID + t];
      cur_grp = grp;
      __syncthreads();
    }
    {
      const half8 *pk = packed;
      asm volatile("" : "+s"(pk));   // opaque per tile (see Tile::dma)
      tl.packed = pk;
    }
    tl.xp = xrow(tile);
    tl.xn = xrow(tile + 1 < t_end ? tile + 1 : tile);
    tl.op = out + ((size_t)tile * TILE + tl.wave * 32 + n) * HID + 4 * tl.half;
#pragma unroll
    for (int ob = 0; ob < 16; ++ob) tl.acc[ob] = f32x16{0.f};
    run_steps<K_IN>(tl, std::make_integer_sequence<int, T::NP>{});
  }
  wait_vmcnt<0>();
  // the last tile's "next tile" prefetch is never consumed: keep its destination
  // registers reserved until the data has landed (the compiler sees the opaque loads
  // as complete at issue and would otherwise reuse the registers under them)
#pragma unroll
  for (int k = 0; k < XS; ++k) asm volatile("" ::"v"(tl.xr[k][0]), "v"(tl.xr[k][1]));
  if ((tl.amax16 & 0xffffu) >= 0x7bffu || (tl.amax16 >> 16) >= 0x7bffu) atomicOr(status, 2u);
}

}  // namespace

RFD_API size_t rfd_resblock_packed_bytes(int k_in) {
  return (size_t)(k_in / 16 + 8) * PIECE_BYTES;
}

RFD_API int rfd_resblock_pack(int k_in, int ld, const float *fc0_w, const float *shortcut_w,
                              const float *fc1_w, int kw0, int kw1, void *packed, void *stream) {
  if (k_in != 256 && k_in != 512) {
    rfd_set_error("rfd_resblock_pack: k_in must be 256 or 512", hipErrorInvalidValue);
    return (int)hipErrorInvalidValue;
  }
  const size_t total = rfd_resblock_packed_bytes(k_in) / sizeof(_Float16);
  hipLaunchKernelGGL(resblock_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, fc0_w, shortcut_w, fc1_w, k_in, ld, kw0, kw1, (_Float16 *)packed,
                     total);
  RFD_CHECK_LAUNCH();
  return 0;
}

RFD_API int rfd_resblock_f16x3(int M, int k_in, int rows_per_group, const float *x, const void *packed,
                               const float *g0, const float *gs, float *out, int kw0, int kw1,
                               void *stream) {
  if (M <= 0) return 0;
  if ((k_in != 256 && k_in != 512) || M % TILE || rows_per_group <= 0 || rows_per_group % TILE) {
    rfd_set_error("rfd_resblock_f16x3: k_in in {256,512}, M and rows_per_group multiples of 128",
                  hipErrorInvalidValue);
    return (int)hipErrorInvalidValue;
  }
  RfdWorkspace *ws;
  int rc = rfd_get_workspace(&ws);
  if (rc) return rc;
  const int n_tiles = M / TILE;
  const int ncu = ws->num_cu > 0 ? ws->num_cu : 256;
  const int tiles_per_wg = ceil_div(n_tiles, ncu);
  const int grid = ceil_div(n_tiles, tiles_per_wg);
  const float inv_s1 = ldexpf(1.f, -kw0), inv_out = ldexpf(1.f, -(KA + kw1));
  hipStream_t s = (hipStream_t)stream;
  if (k_in == 256)
    hipLaunchKernelGGL(resblock_kernel<256>, dim3(grid), dim3(256), 0, s, n_tiles, rows_per_group / TILE, x,
                       (const half8 *)packed, g0, gs, out, inv_s1, inv_out, ws->status, tiles_per_wg);
  else
    hipLaunchKernelGGL(resblock_kernel<512>, dim3(grid), dim3(256), 0, s, n_tiles, rows_per_group / TILE, x,
                       (const half8 *)packed, g0, gs, out, inv_s1, inv_out, ws->status, tiles_per_wg);
  RFD_CHECK_LAUNCH();
  return 0;
}
