// occ_decoder8.hip -- the fused occupancy decoder (see occ_decoder.hip for the network, the
// folding and the arithmetic) with EIGHT waves per workgroup.
//
// Same 128-point tile and the same LDS weight ring, but a wave owns 16 points x all 256
// channels and uses v_mfma_f32_16x16x32_f16: the residual stream is 16 tiles x 4 accumulator
// registers, the block input 8 k-steps x (hi, lo) fragments -- about 220 registers, so TWO waves
// share each SIMD.  While one of them is stuck issuing an LDS-DMA transfer, parked at a wait /
// barrier or doing epilogue arithmetic, the other keeps the matrix pipe busy; the 4-wave kernel
// (one 496-register wave per SIMD) has nobody to fill those slots.
//
// Layouts (16x16x32): A fragment = lane (m = lane & 15, kg = lane >> 4) holds k = 8 kg + j of
// output channel m; B fragment = lane (n = lane & 15, kg) holds k = 8 kg + j of point n;
// D = lane (n, g = lane >> 4) holds channels 4 g + r of a 16-channel tile.  A 32-wide k-step
// covers channel tiles 2 ks and 2 ks + 1, and lane (n, g) feeds its own accumulators back:
// slot j < 4 = tile 2 ks, channel 4 g + j; j >= 4 = tile 2 ks + 1, channel 4 g + (j - 4).  The
// pack kernel orders every weight matrix's columns accordingly, so activations never leave
// the lane (as in the 4-wave kernel).
#include "common.h"
#include <stdlib.h>
#include <type_traits>
#include "../../include/rfd_occ.h"

// This file is the shipped kernel and nothing else.  The timing-only side builds of rounds 2-4 (operands zeroed,
// weight-fragment reads thinned, correction products re-typed, static wave priorities, round 2's two-set prefetch ...:
// results WRONG on purpose) live in tools/ab/dec8_ablation.patch, which tools/ab/build_variants.py applies to a
// scratch copy of this source; tests/test_isa_audit.py checks that the patch still applies and that the patched
// source built without any switch is this kernel, instruction for instruction.
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

constexpr int H = RFD_OCC_HIDDEN;
constexpr int NB = RFD_OCC_BLOCKS;
constexpr int TILE = RFD_OCC_TILE;
constexpr int ROWS = RFD_OCC_TABLE_ROWS;
constexpr int FRAG_HALVES = 64 * 8;
constexpr size_t PACKED_HALVES = (size_t)NB * 8 * 64 * FRAG_HALVES;     // same size as the 4-wave stream
constexpr int SMEM_TAB_BYTES = (ROWS * H + H * 3 + H) * 4;
constexpr int HALF_FRAGS = 32;
constexpr int HALF_BYTES = HALF_FRAGS * FRAG_HALVES * 2;                  // 32 KiB
constexpr int N_HALVES = NB * 8 * 2;
constexpr int SMEM_BYTES = SMEM_TAB_BYTES + 4 * HALF_BYTES;

// ---- chunk claiming (round 4) ----------------------------------------------------------------------------------
// The grid is persistent (one workgroup per CU: a workgroup holds a CU's whole register file and 155 KiB of its LDS),
// so a workgroup whose CU still runs somebody else's waves at launch -- another scene's furthest-point sampling (32
// workgroups for ~5 ms), a fill, the device-to-host blit of the previous scene's meshes -- cannot start until they
// have left.  With the static partition of rounds 1-3 (tiles_per_wg consecutive tiles per workgroup) such a late
// workgroup set the kernel's end: the decoder ran 8 % (64^3) to 15 % (128^3) slower inside a scene than alone.
// Now the tiles are handed out: chunk k of a "factoring" schedule -- batches of W = gridDim.x chunks, the chunks of
// batch j all of size s_j = max(1, ceil(R_j / (2 W))), R_j = tiles not yet handed out before batch j -- is taken
// with one atomicAdd by thread 0 and broadcast through LDS.  The first chunks are large (the first batch is half of the
// launch: few conditioning-table loads), the last are single tiles (the kernel's tail is at most one tile, ~0.1 ms),
// and a workgroup that starts late simply takes fewer.  The range of chunk k is a pure function of (k, n_tiles, W):
// no chunk list, same C ABI.  A chunk may straddle two proposals (the tile loop reloads the table exactly as before);
// results do not depend on who computes a tile, so they are bit-identical to the static partition's.
// Counter pair {next chunk, workgroups done} comes from a pool in the workspace (slot = launch sequence number mod
// RFD_CLAIM_SLOTS, so concurrent launches on different streams never share one) and is reset by the last workgroup
// to leave: no memset launch in front of the kernel.
// cap > 0: no chunk larger than `cap` tiles (the one-chunk-per-workgroup launch below: a workgroup must not hold its CU
// for longer than ~cap x 0.1 ms).
__host__ __device__ __forceinline__ void chunk_range(int k, int n_tiles, int W, int &b, int &e, int cap = 0) {
  int base = 0, rem = n_tiles;
  const int batch = k / W, i = k - batch * W;
  for (int q = 0; q < batch && rem > 0; ++q) {
    int s = (rem + 2 * W - 1) / (2 * W);
    s = s < 1 ? 1 : s;
    s = cap > 0 && s > cap ? cap : s;
    const int take = s * W < rem ? s * W : rem;
    base += take;
    rem -= take;
  }
  int s = (rem + 2 * W - 1) / (2 * W);
  s = s < 1 ? 1 : s;
  s = cap > 0 && s > cap ? cap : s;
  const long long lb = (long long)i * s;
  if (lb >= rem) {
    b = e = n_tiles;
    return;
  }
  b = base + (int)lb;
  e = (lb + s < rem) ? b + s : base + rem;
}

// Stream order = consumption order.  Chunk (blk, mb):
//   fragments  0..31 : fc_0 rows of block mb (tiles 2mb, 2mb+1): q = 4 ks + 2 tt + s
//   fragments 32..63 : fc_1 columns of slab mb (k-step mb) for the 16 output tiles: q = 2 t + s
// in_ch(ks, kg, j) = 32 ks + 16 (j >> 2) + 4 kg + (j & 3).
__global__ void pack8_kernel(const float *__restrict__ fc0_w, const float *__restrict__ fc1_w, int kw0_0,
                             int kw0_1, int kw0_2, int kw0_3, int kw0_4, int kw1,
                             _Float16 *__restrict__ packed) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= PACKED_HALVES) return;
  const int j = e & 7;
  const int lane = (e >> 3) & 63;
  const int frag = (int)(e >> 9);
  const int q = frag & 31;
  const int second = (frag >> 5) & 1;
  const int mb = (frag >> 6) & 7;
  const int blk = frag >> 9;
  const int m = lane & 15, kg = lane >> 4;
  const int s = q & 1;
  int out_ch, in_ch, kw;
  const float *W;
  if (!second) {
    const int ks = q >> 2, tt = (q >> 1) & 1;
    out_ch = 32 * mb + 16 * tt + m;
    in_ch = 32 * ks + 16 * (j >> 2) + 4 * kg + (j & 3);
    W = fc0_w + (size_t)blk * H * H;
    kw = blk == 0 ? kw0_0 : blk == 1 ? kw0_1 : blk == 2 ? kw0_2 : blk == 3 ? kw0_3 : kw0_4;
  } else {
    const int t = q >> 1;
    out_ch = 16 * t + m;
    in_ch = 32 * mb + 16 * (j >> 2) + 4 * kg + (j & 3);
    W = fc1_w + (size_t)blk * H * H;
    kw = kw1;
  }
  const float w = ldexpf(W[(size_t)out_ch * H + in_ch], kw);
  const _Float16 hi = (_Float16)w;
  _Float16 lo = (_Float16)(w - (float)hi);
  packed[e] = s == 0 ? hi : lo;
}

__device__ __forceinline__ f32x4 mfma16(half8 a, half8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ unsigned pk_max_u16(unsigned a, unsigned b) {
  unsigned r;
  asm("v_pk_max_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// relu(s x + t) of two values -> packed f16 hi (round to zero) and lo words
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

// the two channel tiles of one k-step -> B fragment pair; S / T rows at channel 32 ks
template <bool WITH_LO>
__device__ __forceinline__ void act_kstep(const f32x4 &x0, const f32x4 &x1, const float *S, const float *T, int ch,
                                          half8 &hi, half8 &lo, unsigned &amax16) {
  const f32x4 s0 = *reinterpret_cast<const f32x4 *>(S + ch), t0 = *reinterpret_cast<const f32x4 *>(T + ch);
  const f32x4 s1 = *reinterpret_cast<const f32x4 *>(S + ch + 16), t1 = *reinterpret_cast<const f32x4 *>(T + ch + 16);
  unsigned hw[4], lw[4];
  act2<WITH_LO>(x0[0], x0[1], s0[0], s0[1], t0[0], t0[1], hw[0], lw[0], amax16);
  act2<WITH_LO>(x0[2], x0[3], s0[2], s0[3], t0[2], t0[3], hw[1], lw[1], amax16);
  act2<WITH_LO>(x1[0], x1[1], s1[0], s1[1], t1[0], t1[1], hw[2], lw[2], amax16);
  act2<WITH_LO>(x1[2], x1[3], s1[2], s1[3], t1[2], t1[3], hw[3], lw[3], amax16);
  hi = __builtin_bit_cast(half8, u32x4{hw[0], hw[1], hw[2], hw[3]});
  lo = __builtin_bit_cast(half8, u32x4{lw[0], lw[1], lw[2], lw[3]});
}

// wave w moves fragments 4w..4w+3 of a half
__device__ __forceinline__ void dma_piece8(const half8 *__restrict__ packed, unsigned char *s_slots, int h, int j,
                                           int wave, int lane) {
  const int frag = wave * 4 + j;
  const int hs = h >= N_HALVES ? h - N_HALVES : h;
  // wave-uniform part computed apart from the lane part (SGPR pair + one 32-bit lane offset; no reassociation into a
  // per-piece 64-bit VGPR sum chain)
  unsigned long long b64 = (unsigned long long)packed + ((size_t)hs * HALF_FRAGS + frag) * 1024;
  asm("" : "+s"(b64));
  const char *base = reinterpret_cast<const char *>(b64);
  __builtin_amdgcn_global_load_lds((gbl_void *)(base + (unsigned)(lane * 16)),
                                   (lds_void *)(s_slots + (h & 3) * HALF_BYTES + frag * 1024), 16, 0, 0);
}

// ---- weight-fragment prefetch with PROVABLY disjoint destinations --------------------------------------------
// A k-step's four fragments (hi/lo of two channel tiles) are fetched one step ahead.  Round 2 let the compiler
// place them: it reused the registers the previous step's MFMAs had just read as SrcA (0-6 wait states earlier),
// renamed accumulators between MFMAs and sank a slab's last MFMAs below the slab-end barrier.  With unequal static
// wave priorities that build returned wrong 16-point groups (reproduced from the git history in round 3, 10/10 cold
// processes).  profiles/r03_decoder_hazard.txt sections 7-9: the sunk MFMAs / reused source registers are NOT the cause
// (an assembly-level bisect of the failing binary: moving them back or 512 wait states in between change nothing).
// What goes wrong is one VALU move at the END of the tile prologue (fc_p) of the LOW-priority wave: it reads 0 where an
// LDS-loaded weight should be, in lanes 48-63, so ONE channel of H' loses its y term -- every one of 60 dumped wrong
// groups is explained exactly by that (tools/fault_model.py) -- while the wave's SIMD partner, at the higher priority,
// is ~250 instructions ahead in the block-input code's LDS bursts and MFMAs (no barrier in between); whether it
// happens depends on the code layout modulo 32 bytes and on the partner's timing.  Six isolated hardware mechanisms
// are excluded by micro reproducers; the hardware cause is open.
// What is known to hold: no s_setprio -> never wrong (thousands of cold processes); and the structure below, WITH
// the priority put back, is clean at all eight code layouts where round 2's structure fails at three.  So THREE
// sets rotate -- in use (cur), in flight (nxt), last read (prv) -- and the structure, not the allocator's mood,
// keeps them apart:
//   * step_fence(): nothing is scheduled across a k-step boundary or across a barrier;
//   * keep_alive(cur, prv) at the END of every step: both sets stay allocated for the whole step, so neither the
//     prefetch destinations nor any other load issued in step k (conditioning-table reads) can be given a register
//     that the MFMAs of step k or k-1 read.  A destination was therefore last read >= one whole k-step (six
//     MFMAs) earlier.  tools/audit_mfma_war.py checks exactly that on the generated assembly (CPU suite) -- a
//     conservative invariant of the generated code, not the proof of a root cause.
// The loads themselves are plain loads: hipcc counts them (lgkmcnt) and pads the MFMA-SrcC write-after-read states
// itself.  (An asm-issued variant measured 1 % faster in one build and returned WRONG logits in another: hipcc may
// copy an asm load's destination register before the data has landed -- it does not know the load is in flight.)
struct Frag4 {
  half8 h0, l0, h1, l1;
};

__device__ __forceinline__ void keep_alive(const Frag4 &a, const Frag4 &b) {
  asm volatile("" ::"v"(a.h0), "v"(a.l0), "v"(a.h1), "v"(a.l1), "v"(b.h0), "v"(b.l0), "v"(b.h1), "v"(b.l1));
}
__device__ __forceinline__ void keep_alive(const Frag4 &a) {
  asm volatile("" ::"v"(a.h0), "v"(a.l0), "v"(a.h1), "v"(a.l1));
}

template <int OFF>
__device__ __forceinline__ void frag_issue(Frag4 &d, const half8 *base) {
  const half8 *w = base + OFF / 16;
  d.h0 = w[0];
  d.l0 = w[64];
  d.h1 = w[128];
  d.l1 = w[192];
}

// Nothing is scheduled across a k-step boundary (see above).
__device__ __forceinline__ void step_fence() { __builtin_amdgcn_sched_barrier(0); }

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

template <int TERMS>
__global__ __launch_bounds__(512) void occ_decode8_kernel(
    int n_tiles, const float *__restrict__ pts, const int *__restrict__ tile_prop,
    const int *__restrict__ tile_src, const half8 *__restrict__ packed, const float *__restrict__ fc_p_w,
    const float *__restrict__ table, const float *__restrict__ fc_out_w, float fc_out_b,
    float *__restrict__ logits, unsigned *status, int tiles_per_wg, const int *__restrict__ lin,
    float *__restrict__ values, unsigned char *__restrict__ pstate, size_t n_per, unsigned *claim) {
  constexpr bool X3 = TERMS == 3;
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM_BYTES];
  __shared__ int s_claim;
  float *s_tab = reinterpret_cast<float *>(smem);
  float *s_wp = s_tab + ROWS * H;
  float *s_wo = s_wp + H * 3;
  unsigned char *s_slots = smem + SMEM_TAB_BYTES;

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int g4 = 4 * (lane >> 4), n = lane & 15;
  unsigned amax16 = 0u;
  // the two correction products of the three-term scheme
  auto corr_hl = [&](f32x4 &acc, const half8 &wh, const half8 &xl) { acc = mfma16(wh, xl, acc); };      // W_hi x a_lo
  auto corr_lh = [&](f32x4 &acc, const half8 &wl, const half8 &xh) { acc = mfma16(wl, xh, acc); };      // W_lo x a_hi
  // NO static priority in the shipped build, and tests/test_isa_audit.py refuses one: unequal priorities of a SIMD's
  // two waves are the one necessary condition of round 2's wrong 16-point groups that is understood (the rest is code
  // layout and the timing of the partner wave's path: profiles/r03_decoder_hazard.txt sections 7-9), and they buy nothing
  // (+-0.3 %).

  // claim != nullptr: chunks handed out dynamically (see chunk_range); nullptr: rounds 1-3's static partition
  // (RFD_DECODER_STATIC=1, kept as the A/B control)
  // thread index re-derived from the wave id (scalar) and mbcnt in the rare paths, so that threadIdx.x is not one more
  // value held in a vector register (= spilled: the kernel runs at the 256-register limit) for the whole kernel
  // (volatile: recomputed where it is used, never hoisted out of the tile loop and held)
  auto tid = [&]() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return wave * 64 + l;
  };
  auto next_chunk = [&](int &b, int &e) {
    if (tid() == 0) s_claim = (int)atomicAdd(claim, 1u);
    __syncthreads();
    const int k = __builtin_amdgcn_readfirstlane(s_claim);
    __syncthreads();                      // everybody has read s_claim before thread 0 can overwrite it
    chunk_range(k, n_tiles, (int)gridDim.x, b, e);
  };
  int t_begin, t_end;
  if (claim) {
    next_chunk(t_begin, t_end);
  } else if (tiles_per_wg < 0) {
    // one chunk per workgroup (NOT persistent): chunk = blockIdx.x of the capped schedule (W = -tiles_per_wg >> 8,
    // cap = -tiles_per_wg & 255).  The hardware hands the workgroups out as CUs fall free -- the same run-time balance as
    // claiming -- and BETWEEN two of them a CU is up for grabs: another stream's kernel (the next scene's furthest-point
    // sampling, a MISE pass, a GEMM) gets in within one chunk (~1-2 ms) instead of waiting for the whole launch.
    chunk_range((int)blockIdx.x, n_tiles, (-tiles_per_wg) >> 8, t_begin, t_end, (-tiles_per_wg) & 255);
  } else {
    t_begin = blockIdx.x * tiles_per_wg;
    t_end = (t_begin + tiles_per_wg) < n_tiles ? (t_begin + tiles_per_wg) : n_tiles;
  }
  int cur_prop = -1;
  bool ring_primed = false;
  while (t_begin < t_end) {
  for (int tile = t_begin; tile < t_end; ++tile) {
    const int prop = tile_prop[tile];
    if (prop < 0) continue;
    // the weight stream is the same for every tile, so the tail of a tile always fetches the first halves of "the
    // next tile" when chunks are claimed (whichever tile that will be; a workgroup's last prefetch is simply not used)
    const bool has_next = claim != nullptr || tile + 1 < t_end;
    const size_t pidx = (size_t)tile * TILE + wave * 16 + n;
    const size_t sidx = (size_t)(tile_src ? tile_src[tile] : tile) * TILE + wave * 16 + n;
    const float px = pts[sidx * 3 + 0], py = pts[sidx * 3 + 1], pz = pts[sidx * 3 + 2];

    if (!ring_primed) {
#pragma unroll
      for (int j = 0; j < 12; ++j) dma_piece8(packed, s_slots, j >> 2, j & 3, wave, lane);
    }
    if (prop != cur_prop) {
      __syncthreads();
      const f32x4 *src = reinterpret_cast<const f32x4 *>(table + (size_t)prop * ROWS * H);
      f32x4 *dst = reinterpret_cast<f32x4 *>(s_tab);
      for (int i = tid(); i < ROWS * H / 4; i += 512) dst[i] = src[i];
      if (!ring_primed) {
        int tt = tid();
        asm volatile("" : "+v"(tt));     // once per kernel: keep these two per-lane addresses from being hoisted out of
                                         // the tile loop and held (spilled) for its whole duration
        for (int i = tt; i < H * 3; i += 512) s_wp[i] = fc_p_w[i];
        if (tt < H) s_wo[tt] = fc_out_w[tt];
      }
      cur_prop = prop;
    }
    ring_primed = true;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- fc_p (+ fc_z bias): H' = (Wp p + bp + zb) 2^KH
    f32x4 Hs[16];
#pragma unroll
    for (int tt = 0; tt < 16; ++tt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ch = 16 * tt + g4 + r;
        float v = s_tab[ch];
        v = __builtin_fmaf(s_wp[ch * 3 + 0], px, v);
        v = __builtin_fmaf(s_wp[ch * 3 + 1], py, v);
        v = __builtin_fmaf(s_wp[ch * 3 + 2], pz, v);
        Hs[tt][r] = v;
      }
    // All eight waves finish the prologue's LDS reads before any of them enters the block-input code.  Round 2's wrong
    // 16-point groups were exactly this stage going wrong in the LOW-priority wave of a SIMD (one fc_p weight read as
    // 0 in lanes 48-63) while its partner, at a static higher priority, was ~250 instructions ahead in the block-input
    // code's LDS bursts and MFMAs; on the failing binary a barrier here cures it (0/4 against 4/4 for the same bytes
    // without it: profiles/r03_decoder_hazard.txt section 8).  The shipped build has no priorities and never showed
    // the fault, so this is a belt: one barrier per tile, next to ~45 others.
    __syncthreads();      // (the positive control of tools/audit_prologue_lds.py is a side build without it)

    half8 ahi[8], alo[8];
    for (int blk = 0; blk < NB; ++blk) {
      const float *S0 = s_tab + (1 + 4 * blk) * H, *T0 = S0 + H, *S1 = T0 + H, *T1 = S1 + H;
      // ---- block input a' = relu(S0' H' + T0'), fused with fc_0 of output block 0: k-step ks
      // needs only channel tiles 2ks, 2ks+1, so k-step ks+1 is converted under its MFMAs
      f32x4 acc_cur[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
      {
        const half8 *a0 = reinterpret_cast<const half8 *>(s_slots + ((2 * blk * 8) & 3) * HALF_BYTES) + lane;
        Frag4 fs[3];
        frag_issue<0>(fs[0], a0);
        act_kstep<X3>(Hs[0], Hs[1], S0, T0, g4, ahi[0], alo[0], amax16);
        static_for<0, 8>([&](auto kc) {
          constexpr int ks = decltype(kc)::value;
          Frag4 &cur = fs[ks % 3], &nxt = fs[(ks + 1) % 3], &prv = fs[(ks + 2) % 3];
          step_fence();
          if constexpr (ks < 7) {
            frag_issue<4096 * (ks + 1)>(nxt, a0);
          }
          if constexpr (ks < 7)
            act_kstep<X3>(Hs[2 * ks + 2], Hs[2 * ks + 3], S0, T0, 32 * (ks + 1) + g4, ahi[ks + 1], alo[ks + 1], amax16);
          acc_cur[0] = mfma16(cur.h0, ahi[ks], acc_cur[0]);
          acc_cur[1] = mfma16(cur.h1, ahi[ks], acc_cur[1]);
          if (X3) {
            corr_hl(acc_cur[0], cur.h0, alo[ks]);
            corr_hl(acc_cur[1], cur.h1, alo[ks]);
            corr_lh(acc_cur[0], cur.l0, ahi[ks]);
            corr_lh(acc_cur[1], cur.l1, ahi[ks]);
          }
          if constexpr (ks == 0) keep_alive(cur);
          else keep_alive(cur, prv);
        });
      }
      step_fence();
      __syncthreads();

      for (int mb = 0; mb < 8; ++mb) {
        const int c = blk * 8 + mb;
        // ---- epilogue of fc_0 block mb -> a2' (the B operand of fc_1's k-slab mb)
        f32x4 acc_next[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        half8 bhi, blo;
        act_kstep<X3>(acc_cur[0], acc_cur[1], S1, T1, 32 * mb + g4, bhi, blo, amax16);
        auto issue_dma = [&](int j) {          // piece j (0..7) of this slab's two halves
          const int h = 2 * c + 3 + (j >> 2);
          if (h < N_HALVES || has_next) dma_piece8(packed, s_slots, h, j & 3, wave, lane);
        };
        // ---- phase A: fc_0 block mb+1 (two accumulator chains) with this iteration's LDS-DMA pieces in
        // between; phase B: H'[t] += fc_1[16t.., slab mb] a2', sixteen accumulators, two chains at a time.
        // ONE sequence of 16 k-steps for the fragment rotation: step 7 of phase A prefetches phase B's first set.
        const half8 *aA = reinterpret_cast<const half8 *>(s_slots + ((2 * c + 2) & 3) * HALF_BYTES) + lane;
        const half8 *aB = reinterpret_cast<const half8 *>(s_slots + ((2 * c + 1) & 3) * HALF_BYTES) + lane;
        Frag4 fs[3];
        if (mb < 7) {
          frag_issue<0>(fs[0], aA);
          static_for<0, 8>([&](auto kc) {
            constexpr int ks = decltype(kc)::value;
            Frag4 &cur = fs[ks % 3], &nxt = fs[(ks + 1) % 3], &prv = fs[(ks + 2) % 3];
            step_fence();
            if constexpr (ks < 7) {
              frag_issue<4096 * (ks + 1)>(nxt, aA);
            } else {
              frag_issue<0>(nxt, aB);
            }
            issue_dma(ks);
            acc_next[0] = mfma16(cur.h0, ahi[ks], acc_next[0]);
            acc_next[1] = mfma16(cur.h1, ahi[ks], acc_next[1]);
            if (X3) {
              corr_hl(acc_next[0], cur.h0, alo[ks]);
              corr_hl(acc_next[1], cur.h1, alo[ks]);
              corr_lh(acc_next[0], cur.l0, ahi[ks]);
              corr_lh(acc_next[1], cur.l1, ahi[ks]);
            }
            if constexpr (ks == 0) keep_alive(cur);
            else keep_alive(cur, prv);
          });
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) issue_dma(j);
          frag_issue<0>(fs[8 % 3], aB);
        }
        static_for<0, 8>([&](auto kc) {
          constexpr int tp = decltype(kc)::value;
          constexpr int ks = 8 + tp;
          Frag4 &cur = fs[ks % 3], &nxt = fs[(ks + 1) % 3], &prv = fs[(ks + 2) % 3];
          step_fence();
          if constexpr (tp < 7) {
            frag_issue<4096 * (tp + 1)>(nxt, aB);
          }
          Hs[2 * tp] = mfma16(cur.h0, bhi, Hs[2 * tp]);
          Hs[2 * tp + 1] = mfma16(cur.h1, bhi, Hs[2 * tp + 1]);
          if (X3) {
            corr_hl(Hs[2 * tp], cur.h0, blo);
            corr_hl(Hs[2 * tp + 1], cur.h1, blo);
            corr_lh(Hs[2 * tp], cur.l0, bhi);
            corr_lh(Hs[2 * tp + 1], cur.l1, bhi);
          }
          // tp == 0: the set before this one is phase A's step 7 (mb < 7) or nothing (mb == 7, behind a barrier)
          if constexpr (tp == 0) {
            if (mb < 7) keep_alive(cur, prv);
            else keep_alive(cur);
          } else {
            keep_alive(cur, prv);
          }
        });
        // (running the two phases in opposite order on the two waves of a SIMD, spreading the DMA
        // pieces over both phases, or letting one wave of a pair issue all of them: all within 0.7 %,
        // the swap -8 % with 28 spilled registers -- profiles/r02_decoder_ablation.txt)
        step_fence();       // the slab's last MFMAs stay in front of the barrier (hipcc sinks them behind it otherwise,
                            // right in front of the next slab's conditioning-table loads into the registers they read)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        acc_cur[0] = acc_next[0];
        acc_cur[1] = acc_next[1];
      }
    }

    // ---- out = fc_out(relu(CBN_f(h)))
    const float *Sf = s_tab + 21 * H, *Tf = Sf + H;
    float part = 0.f;
#pragma unroll
    for (int tt = 0; tt < 16; ++tt) {
      const int ch0 = 16 * tt + g4;
      const f32x4 s4 = *reinterpret_cast<const f32x4 *>(Sf + ch0);
      const f32x4 t4 = *reinterpret_cast<const f32x4 *>(Tf + ch0);
      const f32x4 w4 = *reinterpret_cast<const f32x4 *>(s_wo + ch0);
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
        // MISE update fused into the epilogue (mise.pyx:101-102: value stored, point known): the
        // logits never take the detour through a tile-ordered buffer and a scatter launch
        const int l = lin[sidx];
        if (l >= 0) {
          values[(size_t)prop * n_per + l] = part + fc_out_b;
          pstate[(size_t)prop * n_per + l] = 2;
        }
      } else {
        size_t po = pidx;
        asm volatile("" : "+v"(po));      // once per tile: no per-lane store address held (spilled) across the whole tile loop
        logits[po] = part + fc_out_b;
      }
    }
  }
    if (!claim) break;
    next_chunk(t_begin, t_end);
  }
  if ((amax16 & 0xffffu) >= 0x7bffu || (amax16 >> 16) >= 0x7bffu) atomicOr(status, 2u);
  if (claim) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the last tile's unused ring prefetch has landed
    if (tid() == 0) {
      // every workgroup's last (failed) claim precedes its increment of claim[1]: the last one out resets the pair
      __threadfence();
      if (atomicAdd(claim + 1, 1u) == gridDim.x - 1) {
        __atomic_store_n(claim, 0u, __ATOMIC_RELAXED);
        __atomic_store_n(claim + 1, 0u, __ATOMIC_RELAXED);
      }
    }
  }
}

}  // namespace

// number of non-empty chunks of the capped schedule
static int chunk_count(int n_tiles, int W, int cap) {
  int rem = n_tiles, n = 0;
  while (rem > 0) {
    int s = (rem + 2 * W - 1) / (2 * W);
    s = s < 1 ? 1 : s;
    s = cap > 0 && s > cap ? cap : s;
    const int take = s * W < rem ? s * W : rem;
    n += (take + s - 1) / s;
    rem -= take;
  }
  return n;
}

// The chunk schedule of the eight-wave decoder as the host sees it (tests/test_chunk_schedule.py: the chunks of a launch
// partition [0, n_tiles) for every grid size).
RFD_API int rfd_occ_chunk_range(int k, int n_tiles, int n_workgroups, int *begin, int *end) {
  if (k < 0 || n_tiles < 0 || n_workgroups <= 0 || !begin || !end) return (int)hipErrorInvalidValue;
  chunk_range(k, n_tiles, n_workgroups, *begin, *end);
  return 0;
}

// The same with chunks capped at max_chunk tiles (the one-chunk-per-workgroup launch); returns the number of chunks
// through *n_chunks when it is not NULL.
RFD_API int rfd_occ_chunk_range_capped(int k, int n_tiles, int n_workgroups, int max_chunk, int *begin, int *end,
                                       int *n_chunks) {
  if (k < 0 || n_tiles < 0 || n_workgroups <= 0 || max_chunk < 0 || !begin || !end) return (int)hipErrorInvalidValue;
  chunk_range(k, n_tiles, n_workgroups, *begin, *end, max_chunk);
  if (n_chunks) *n_chunks = chunk_count(n_tiles, n_workgroups, max_chunk);
  return 0;
}

RFD_API int rfd_occ_pack_weights_w8(const float *fc0_w, const float *fc1_w, const int *kw0, int kw1,
                                    void *packed, void *stream) {
  const int threads = 256;
  const int blocks = (int)((PACKED_HALVES + threads - 1) / threads);
  hipLaunchKernelGGL(pack8_kernel, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, fc0_w, fc1_w, kw0[0],
                     kw0[1], kw0[2], kw0[3], kw0[4], kw1, (_Float16 *)packed);
  RFD_CHECK_LAUNCH();
  return 0;
}

// Launch shape of the eight-wave decoder.  The default -- persistent grid, one workgroup per CU, tiles claimed at run
// time -- is what ships; the other shapes are the A/B controls the tests and profiles/r04_*.txt compare it with
// (bit-identical results, all measured slower): static_partition = rounds 1-3's tiles_per_wg consecutive tiles per
// workgroup; cus = n: a persistent grid of n < num_cu workgroups; chunk_cap = c (1..255): NOT persistent, one
// workgroup per chunk of at most c tiles.  Initialised ONCE from RFD_DECODER_STATIC / RFD_DECODER_CUS /
// RFD_DECODER_CHUNK (no getenv on the launch path), changed at run time through rfd_occ_set_launch_shape.
//
// tail_tiles = n: a launch of at most n tiles (and the default shape otherwise) goes to the one-wave, no-LDS kernel of
// occ_decoder_tail.hip, which does not need an empty CU to start (bit-identical logits); 0 = never.  RFD_DECODER_TAIL_TILES /
// rfd_occ_set_tail_tiles.
int rfd_occ_tail_launch(int n_tiles, const float *pts, const int *tile_prop, const int *tile_src, const void *packed,
                        const float *fc_p_w, const float *table, const float *fc_out_w, float fc_out_b, float *logits,
                        unsigned *status, const int *lin, float *values, unsigned char *pstate, size_t n_per, int mode,
                        hipStream_t stream);
namespace {
constexpr int TAIL_TILES_DEFAULT = 384;
struct LaunchShape {
  std::atomic<int> static_partition, cus, chunk_cap, tail_tiles;
  LaunchShape() {
    const char *e;
    static_partition.store((e = getenv("RFD_DECODER_STATIC")) ? (atoi(e) != 0) : 0);
    cus.store((e = getenv("RFD_DECODER_CUS")) ? atoi(e) : 0);
    chunk_cap.store((e = getenv("RFD_DECODER_CHUNK")) ? atoi(e) : 0);
    tail_tiles.store((e = getenv("RFD_DECODER_TAIL_TILES")) ? atoi(e) : TAIL_TILES_DEFAULT);
  }
};
LaunchShape &launch_shape() {
  static LaunchShape ls;
  return ls;
}
}  // namespace

// Each argument: >= 0 sets, < 0 leaves unchanged.  Returns 0.
RFD_API int rfd_occ_set_launch_shape(int static_partition, int cus, int chunk_cap) {
  LaunchShape &ls = launch_shape();
  if (static_partition >= 0) ls.static_partition.store(static_partition != 0);
  if (cus >= 0) ls.cus.store(cus);
  if (chunk_cap >= 0) ls.chunk_cap.store(chunk_cap > 255 ? 255 : chunk_cap);
  return 0;
}

// Launches of at most n_tiles tiles use the small-footprint kernel (0 = never); n_tiles < 0 leaves the setting.  Returns
// the previous value.
RFD_API int rfd_occ_set_tail_tiles(int n_tiles) {
  LaunchShape &ls = launch_shape();
  const int old = ls.tail_tiles.load();
  if (n_tiles >= 0) ls.tail_tiles.store(n_tiles);
  return old;
}

static int decode_w8(int n_tiles, const float *pts, const int *tile_prop, const int *tile_src,
                     const void *packed, const float *fc_p_w, const float *table, const float *fc_out_w,
                     float fc_out_b, float *logits, const int *lin, float *values, unsigned char *pstate,
                     size_t n_per, int mode, void *stream) {
  if (n_tiles <= 0) return 0;
  RfdWorkspace *ws;
  int rc = rfd_get_workspace(&ws);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  int ncu = ws->num_cu > 0 ? ws->num_cu : 256;
  const LaunchShape &ls = launch_shape();
  const int cu_limit = ls.cus.load(std::memory_order_relaxed);
  if (cu_limit > 0 && cu_limit < ncu) ncu = cu_limit;
  const bool static_part = ls.static_partition.load(std::memory_order_relaxed) != 0;
  int cap = ls.chunk_cap.load(std::memory_order_relaxed);
  cap = cap < 0 ? 0 : cap > 255 ? 255 : cap;
  if (!static_part && cap == 0 && cu_limit <= 0 && n_tiles <= ls.tail_tiles.load(std::memory_order_relaxed))
    return rfd_occ_tail_launch(n_tiles, pts, tile_prop, tile_src, packed, fc_p_w, table, fc_out_w, fc_out_b, logits,
                               rfd_status_word(ws, s), lin, values, pstate, n_per, mode, s);
  int tiles_per_wg = ceil_div(n_tiles, ncu);
  int grid = static_part ? ceil_div(n_tiles, tiles_per_wg) : (n_tiles < ncu ? n_tiles : ncu);
  // the counter pair of THIS stream (serial launches: never shared with a launch in flight)
  unsigned *claim = static_part ? nullptr : rfd_claim_pair(ws, s);
  if (!static_part && cap > 0) {
    claim = nullptr;
    tiles_per_wg = -((ncu << 8) | cap);
    grid = chunk_count(n_tiles, ncu, cap);
  }
  if (mode == RFD_OCC_MODE_F16X3) {
    hipLaunchKernelGGL(occ_decode8_kernel<3>, dim3(grid), dim3(512), 0, s, n_tiles, pts, tile_prop, tile_src,
                       (const half8 *)packed, fc_p_w, table, fc_out_w, fc_out_b, logits, rfd_status_word(ws, s), tiles_per_wg,
                       lin, values, pstate, n_per, claim);
  } else if (mode == RFD_OCC_MODE_F16X1) {
    hipLaunchKernelGGL(occ_decode8_kernel<1>, dim3(grid), dim3(512), 0, s, n_tiles, pts, tile_prop, tile_src,
                       (const half8 *)packed, fc_p_w, table, fc_out_w, fc_out_b, logits, rfd_status_word(ws, s), tiles_per_wg,
                       lin, values, pstate, n_per, claim);
  } else {
    rfd_set_error("rfd_occ_decode_w8: unknown mode", hipErrorInvalidValue);
    return (int)hipErrorInvalidValue;
  }
  RFD_CHECK_LAUNCH();
  return 0;
}

RFD_API int rfd_occ_decode_w8(int n_tiles, const float *pts, const int *tile_prop, const int *tile_src,
                              const void *packed, const float *fc_p_w, const float *table,
                              const float *fc_out_w, float fc_out_b, float *logits, int mode, void *stream) {
  return decode_w8(n_tiles, pts, tile_prop, tile_src, packed, fc_p_w, table, fc_out_w, fc_out_b, logits, nullptr,
                   nullptr, nullptr, 0, mode, stream);
}

// Decode + MISE update in one launch: the logit of query slot s (= source slot when tile_src is given) goes to
// values[prop][lin[s]] and marks pstate[prop][lin[s]] = 2 (known); slots with lin < 0 are padding.
RFD_API int rfd_occ_decode_scatter_w8(int n_tiles, const float *pts, const int *tile_prop, const int *tile_src,
                                      const void *packed, const float *fc_p_w, const float *table,
                                      const float *fc_out_w, float fc_out_b, const int *lin, float *values,
                                      unsigned char *pstate, long long n_per, int mode, void *stream) {
  if (!lin || !values || !pstate || n_per <= 0) {
    rfd_set_error("rfd_occ_decode_scatter_w8: lin / values / pstate / n_per", hipErrorInvalidValue);
    return (int)hipErrorInvalidValue;
  }
  return decode_w8(n_tiles, pts, tile_prop, tile_src, packed, fc_p_w, table, fc_out_w, fc_out_b, nullptr, lin,
                   values, pstate, (size_t)n_per, mode, stream);
}
