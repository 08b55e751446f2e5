// occ_decoder_tail.hip -- the occupancy decoder for TAIL launches (a MISE round with a few thousand query points left).
//
// occ_decode8_kernel (occ_decoder8.hip) is built for millions of points: a workgroup owns a CU -- 155 KiB of LDS, the whole
// register file -- and cannot start on a CU that holds a single wave of anybody else.  With several scenes in flight a
// launch of a few hundred tiles therefore WAITS, for the other scenes' marching cubes / subdivision / GEMM workgroups to
// drain, 5-10x longer than it computes (profiles/r05_mise128_rounds.txt section 2; BENCH_r05 per_round).  This kernel has
// the opposite shape: ONE wave per workgroup, 16 query points, no LDS, no barrier, at most 256 vector registers -- it
// fits on any SIMD with half its register file free, next to whatever runs there.  The weight fragments come straight
// from L2 (the 2.6-MB packed stream of rfd_occ_pack_weights_w8, three k-steps ahead in registers), the conditioning table
// from global memory.  Sixteen points per 2.6 MB of weights is L2-bandwidth bound (64 B per clock and CU): 2048 groups take
// 0.21 ms where the main kernel takes 0.12 ms alone -- this kernel is for launches that would otherwise WAIT.  A 16-point group whose slots are all padding (lin < 0) leaves at once: a tail round has ~18 real
// points per proposal in 128-slot tiles.
//
// Arithmetic: every accumulator sees the SAME sequence of matrix instructions as in occ_decode8_kernel (per k-step: W_hi a_hi,
// W_hi a_lo, W_lo a_hi; k-steps in order), the same fc_p / CBN / fc_out code around them -- the logits are bit-identical
// (tests/test_gpu_decoder.py asserts it).  The helpers below are copies of occ_decoder8.hip's: that file is frozen
// (tests/test_isa_audit.py pins its instruction stream) and stays untouched.
//
// Two waves of this kernel share a SIMD and nothing orders them, so one wave's prologue (fc_p) runs beside the other's matrix
// instructions: the first build of this file returned wrong 16-point groups until the library was compiled with
// -fno-slp-vectorize -- hipcc had packed fc_p's y term into v_pk_fma_f32 ... op_sel:[0,1,0], which is wrong in lanes 48-63
// beside another wave's v_mfma on gfx950 (profiles/r06_pk_f32_hazard.txt, tools/hazard/; tests/test_isa_audit.py keeps the
// form out of every kernel).
#include "common.h"
#include "../../include/rfd_occ.h"

int rfd_occ_tail_launch(int n_tiles, const float *pts, const int *tile_prop, const int *tile_src, const void *packed,
                        const float *fc_p_w, const float *table, const float *fc_out_w, float fc_out_b, float *logits,
                        unsigned *status, const int *lin, float *values, unsigned char *pstate, size_t n_per, int mode,
                        hipStream_t stream);

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int H = RFD_OCC_HIDDEN;
constexpr int NB = RFD_OCC_BLOCKS;
constexpr int TILE = RFD_OCC_TILE;
constexpr int ROWS = RFD_OCC_TABLE_ROWS;
constexpr int HALF_FRAGS = 32;                 // fragments (1 KiB each) per half of the packed stream
constexpr int DEPTH = 3;                       // k-steps of weight fragments in flight ahead of the one in use
constexpr int SETS = DEPTH + 1;

__device__ __forceinline__ f32x4 mfma16(half8 a, half8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ unsigned pk_max_u16(unsigned a, unsigned b) {
  unsigned r;
  asm("v_pk_max_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// relu(s x + t) of two values -> packed f16 hi (round to zero) and lo words (occ_decoder8.hip act2)
template <bool WITH_LO>
__device__ __forceinline__ void act2(float x0, float x1, float s0, float s1, float t0, float t1, unsigned &hiw,
                                     unsigned &low, unsigned &amax16) {
  float a0 = __builtin_fmaf(s0, x0, t0), a1 = __builtin_fmaf(s1, x1, t1);
  a0 = a0 > 0.f ? a0 : 0.f;
  a1 = a1 > 0.f ? a1 : 0.f;
  hiw = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a0, a1));
  amax16 = pk_max_u16(amax16, hiw);
  if (WITH_LO) {
    float r0, r1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hiw), "v"(a0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hiw), "v"(a1));
    low = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(r0, r1));
  } else {
    low = 0u;
  }
}

// the S / T values of one k-step's two channel tiles
struct ST {
  f32x4 s0, t0, s1, t1;
};
__device__ __forceinline__ void st_issue(ST &d, const float *S, const float *T, int ch) {
  d.s0 = *reinterpret_cast<const f32x4 *>(S + ch);
  d.t0 = *reinterpret_cast<const f32x4 *>(T + ch);
  d.s1 = *reinterpret_cast<const f32x4 *>(S + ch + 16);
  d.t1 = *reinterpret_cast<const f32x4 *>(T + ch + 16);
}

template <bool WITH_LO>
__device__ __forceinline__ void act_kstep(const f32x4 &x0, const f32x4 &x1, const ST &c, half8 &hi, half8 &lo,
                                          unsigned &amax16) {
  unsigned hw[4], lw[4];
  act2<WITH_LO>(x0[0], x0[1], c.s0[0], c.s0[1], c.t0[0], c.t0[1], hw[0], lw[0], amax16);
  act2<WITH_LO>(x0[2], x0[3], c.s0[2], c.s0[3], c.t0[2], c.t0[3], hw[1], lw[1], amax16);
  act2<WITH_LO>(x1[0], x1[1], c.s1[0], c.s1[1], c.t1[0], c.t1[1], hw[2], lw[2], amax16);
  act2<WITH_LO>(x1[2], x1[3], c.s1[2], c.s1[3], c.t1[2], c.t1[3], hw[3], lw[3], amax16);
  hi = __builtin_bit_cast(half8, u32x4{hw[0], hw[1], hw[2], hw[3]});
  lo = __builtin_bit_cast(half8, u32x4{lw[0], lw[1], lw[2], lw[3]});
}

struct Frag4 {
  half8 h0, l0, h1, l1;
};
// k-step ks of half h: fragments 4 ks .. 4 ks + 3 = (hi, lo) of two channel tiles, 4 KiB in one piece
__device__ __forceinline__ void frag_issue(Frag4 &d, const half8 *half_base, int ks) {
  const half8 *w = half_base + ks * 256;
  d.h0 = w[0];
  d.l0 = w[64];
  d.h1 = w[128];
  d.l1 = w[192];
}

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

template <int TERMS>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void occ_decode_tail_kernel(
    const float *__restrict__ pts, const int *__restrict__ tile_prop, const int *__restrict__ tile_src,
    const half8 *__restrict__ packed, const float *__restrict__ fc_p_w, const float *__restrict__ table,
    const float *__restrict__ fc_out_w, float fc_out_b, float *__restrict__ logits, unsigned *status,
    const int *__restrict__ lin, float *__restrict__ values, unsigned char *__restrict__ pstate, size_t n_per, int n_tiles) {
  constexpr bool X3 = TERMS == 3;
  const int lane = threadIdx.x;
  // group-major: workgroups 0 .. n_tiles - 1 are the FIRST 16 slots of every tile, the next n_tiles the second ...  The tile
  // builder packs a proposal's real points at the front of its tiles, so the groups with work come first and consecutive --
  // round robin over the eight XCDs and their CUs -- and the all-padding groups behind them leave at once.  (Tile-major put
  // the two live groups of every tile on XCDs 0 and 1: eight waves on each of their CUs, none on the other 192.)
  const int grp = blockIdx.x / n_tiles, tile = blockIdx.x - grp * n_tiles, w16 = grp * 16;
  const int g4 = 4 * (lane >> 4), n = lane & 15;
  const int prop = tile_prop[tile];
  if (prop < 0) return;
  const size_t pidx = (size_t)tile * TILE + w16 + n;
  const size_t sidx = (size_t)(tile_src ? tile_src[tile] : tile) * TILE + w16 + n;
  int l = 0;
  if (lin) {
    l = lin[sidx];
    if (__ballot(l >= 0) == 0ull) return;        // sixteen padding slots
  }
  const float px = pts[sidx * 3 + 0], py = pts[sidx * 3 + 1], pz = pts[sidx * 3 + 2];
  const float *tab = table + (size_t)prop * ROWS * H;
  const half8 *wl = packed + lane;               // fragment f of half h: wl[(h * 32 + f) * 64]
  auto half_ptr = [&](int h) { return wl + (size_t)h * HALF_FRAGS * 64; };
  unsigned amax16 = 0u;

  // the weight stream, DEPTH k-steps ahead.  Every stage below is eight k-steps of ONE half and DEPTH < 8, so step k of a
  // stage fetches step k + DEPTH of the same half or step k + DEPTH - 8 of the next stage's half; the rotation (set =
  // step mod SETS, 8 mod SETS = 0) carries over stage boundaries and loop iterations.
  Frag4 fs[SETS];
  {
    const half8 *h0 = half_ptr(0);
    static_for<0, DEPTH>([&](auto kc) { frag_issue(fs[decltype(kc)::value], h0, decltype(kc)::value); });
  }
  auto fetch = [&](auto kc, const half8 *cur, const half8 *next) {
    constexpr int ks = decltype(kc)::value;
    __builtin_amdgcn_sched_barrier(0);           // a k-step's fetch and its matrix instructions stay one unit: the loads of step
                                                 // k + DEPTH go out before step k's first MFMA, the only wait is for step k's set
    if constexpr (ks + DEPTH < 8) frag_issue(fs[(ks + DEPTH) % SETS], cur, ks + DEPTH);
    else frag_issue(fs[(ks + DEPTH) % SETS], next, ks + DEPTH - 8);
  };
  auto mma = [&](f32x4 &a0, f32x4 &a1, const Frag4 &f, const half8 &xh, const half8 &xl) {
    a0 = mfma16(f.h0, xh, a0);
    a1 = mfma16(f.h1, xh, a1);
    if (X3) {
      a0 = mfma16(f.h0, xl, a0);
      a1 = mfma16(f.h1, xl, a1);
      a0 = mfma16(f.l0, xh, a0);
      a1 = mfma16(f.l1, xh, a1);
    }
  };

  // ---- fc_p (+ fc_z bias): H' = (Wp p + bp + zb) 2^KH
  f32x4 Hs[16];
#pragma unroll
  for (int tt = 0; tt < 16; ++tt) {
    const int ch = 16 * tt + g4;
    const f32x4 b = *reinterpret_cast<const f32x4 *>(tab + ch);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = b[r];
      v = __builtin_fmaf(fc_p_w[(ch + r) * 3 + 0], px, v);
      v = __builtin_fmaf(fc_p_w[(ch + r) * 3 + 1], py, v);
      v = __builtin_fmaf(fc_p_w[(ch + r) * 3 + 2], pz, v);
      Hs[tt][r] = v;
    }
  }

  half8 ahi[8], alo[8];
  for (int blk = 0; blk < NB; ++blk) {
    const float *S0 = tab + (1 + 4 * blk) * H, *T0 = S0 + H, *S1 = T0 + H, *T1 = S1 + H;
    // ---- block input a' = relu(S0' H' + T0'), fused with fc_0 of output block 0
    f32x4 acc_cur[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    {
      const half8 *cur = half_ptr(2 * blk * 8), *next = half_ptr(2 * blk * 8 + 2);
      ST st[2];
      st_issue(st[0], S0, T0, g4);
      static_for<0, 8>([&](auto kc) {
        constexpr int ks = decltype(kc)::value;
        fetch(kc, cur, next);
        if constexpr (ks < 7) st_issue(st[(ks + 1) & 1], S0, T0, 32 * (ks + 1) + g4);
        act_kstep<X3>(Hs[2 * ks], Hs[2 * ks + 1], st[ks & 1], ahi[ks], alo[ks], amax16);
        mma(acc_cur[0], acc_cur[1], fs[ks % SETS], ahi[ks], alo[ks]);
      });
    }
    ST sb;
    st_issue(sb, S1, T1, g4);
    for (int mb = 0; mb < 8; ++mb) {
      const int c = blk * 8 + mb;
      // ---- epilogue of fc_0 block mb -> a2' (the B operand of fc_1's k-slab mb)
      f32x4 acc_next[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
      half8 bhi, blo;
      act_kstep<X3>(acc_cur[0], acc_cur[1], sb, bhi, blo, amax16);
      if (mb < 7) st_issue(sb, S1, T1, 32 * (mb + 1) + g4);
      const half8 *hA = half_ptr(2 * c + 2), *hB = half_ptr(2 * c + 1);
      // the half after phase B: phase A of mb + 1, or (mb == 6) phase B of mb 7, or the next block's input stage, or
      // (last slab of the last block) anything mapped -- that prefetch is never used
      const half8 *hN = mb < 6 ? half_ptr(2 * c + 4) : mb == 6 ? half_ptr(2 * c + 3) : blk + 1 < NB ? half_ptr(2 * c + 2) : wl;
      if (mb < 7) {
        // ---- phase A: fc_0 block mb + 1
        static_for<0, 8>([&](auto kc) {
          constexpr int ks = decltype(kc)::value;
          fetch(kc, hA, hB);
          mma(acc_next[0], acc_next[1], fs[ks % SETS], ahi[ks], alo[ks]);
        });
      }
      // ---- phase B: H'[t] += fc_1[16t.., slab mb] a2'
      static_for<0, 8>([&](auto kc) {
        constexpr int tp = decltype(kc)::value;
        fetch(kc, hB, hN);
        mma(Hs[2 * tp], Hs[2 * tp + 1], fs[tp % SETS], bhi, blo);
      });
      acc_cur[0] = acc_next[0];
      acc_cur[1] = acc_next[1];
    }
  }

  // ---- out = fc_out(relu(CBN_f(h)))
  const float *Sf = tab + 21 * H, *Tf = Sf + H;
  float part = 0.f;
#pragma unroll
  for (int tt = 0; tt < 16; ++tt) {
    const int ch0 = 16 * tt + g4;
    const f32x4 s4 = *reinterpret_cast<const f32x4 *>(Sf + ch0);
    const f32x4 t4 = *reinterpret_cast<const f32x4 *>(Tf + ch0);
    const f32x4 w4 = *reinterpret_cast<const f32x4 *>(fc_out_w + ch0);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float a = __builtin_fmaf(s4[e], Hs[tt][e], t4[e]);
      a = a > 0.f ? a : 0.f;
      part = __builtin_fmaf(w4[e], a, part);
    }
  }
  part += __shfl_xor(part, 16);
  part += __shfl_xor(part, 32);
  if (lane < 16) {
    if (lin) {
      if (l >= 0) {
        values[(size_t)prop * n_per + l] = part + fc_out_b;
        pstate[(size_t)prop * n_per + l] = 2;
      }
    } else {
      logits[pidx] = part + fc_out_b;
    }
  }
  if ((amax16 & 0xffffu) >= 0x7bffu || (amax16 >> 16) >= 0x7bffu) atomicOr(status, 2u);
}

}  // namespace

int rfd_occ_tail_launch(int n_tiles, const float *pts, const int *tile_prop, const int *tile_src, const void *packed,
                        const float *fc_p_w, const float *table, const float *fc_out_w, float fc_out_b, float *logits,
                        unsigned *status, const int *lin, float *values, unsigned char *pstate, size_t n_per, int mode,
                        hipStream_t stream) {
  const dim3 grid(n_tiles * (TILE / 16)), block(64);
  if (mode == RFD_OCC_MODE_F16X3) {
    hipLaunchKernelGGL(occ_decode_tail_kernel<3>, grid, block, 0, stream, pts, tile_prop, tile_src, (const half8 *)packed,
                       fc_p_w, table, fc_out_w, fc_out_b, logits, status, lin, values, pstate, n_per, n_tiles);
  } else if (mode == RFD_OCC_MODE_F16X1) {
    hipLaunchKernelGGL(occ_decode_tail_kernel<1>, grid, block, 0, stream, pts, tile_prop, tile_src, (const half8 *)packed,
                       fc_p_w, table, fc_out_w, fc_out_b, logits, status, lin, values, pstate, n_per, n_tiles);
  } else {
    rfd_set_error("rfd_occ_decode_w8: unknown mode", hipErrorInvalidValue);
    return (int)hipErrorInvalidValue;
  }
  RFD_CHECK_LAUNCH();
  return 0;
}
