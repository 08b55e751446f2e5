// Reproducer for the round-2 decoder hazard (profiles/r02_decoder_ablation.txt section 5, profiles/r03_decoder_hazard.txt):
// can data returning from LDS (or a VALU write) into a register that an ALREADY ISSUED v_mfma still has to read
// corrupt that MFMA, and does it take the wave's SIMD partner out-prioritising it?
// Two waves share each SIMD of a 512-thread workgroup (wave w and w + 4).
//   aggressor waves 4-7: an endless stream of v_mfma_f32_16x16x32_f16 on two accumulator chains (the decoder's shape)
//   victim    waves 0-3: six MFMAs that read register set R, GAP filler MFMAs on other registers, NWAIT wait states,
//                        then an overwrite of R with different data; every MFMA result is checked against the OLD R.
//     R     = SrcA / SrcB / SrcC of the six MFMAs
//     DEP 0 = six independent MFMAs (distinct destinations)
//     DEP 1 = the decoder's shape: two accumulator chains, each MFMA takes the result of the one two earlier as
//             SrcC, so the later ones wait in the matrix pipe for their producer -- and for the partner's MFMAs
//     the overwrite is a ds_read_b128 (asynchronous: lands when LDS returns it)
//     BAR 1 = every iteration starts behind an s_barrier that all eight waves take (the decoder's slab boundary)
//   PRIO 0: all waves equal   1: aggressors raised (s_setprio 3)
// A non-zero "bad" count = the distance was not enough under that arrangement.
// Build: hipcc --offload-arch=gfx950 -O2 -o rfdnet_amd/lib/micro/mfma_war tools/micro/mfma_war.hip ; run: mfma_war [iters] [bcast 0|1]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define M_(D, A, B, C) "v_mfma_f32_16x16x32_f16 " D ", " A ", " B ", " C "\n"
#define INDEP(A, B, C) M_("%[d0]", A, B, C) M_("%[d1]", A, B, C) M_("%[d2]", A, B, C) M_("%[d3]", A, B, C) \
                       M_("%[d4]", A, B, C) M_("%[d5]", A, B, C)
#define CHAIN(A, B)    M_("%[d0]", A, B, "0") M_("%[d1]", A, B, "0") M_("%[d0]", A, B, "%[d0]") \
                       M_("%[d1]", A, B, "%[d1]") M_("%[d0]", A, B, "%[d0]") M_("%[d1]", A, B, "%[d1]")
#define FILL ".rept %c[gap]\n v_mfma_f32_16x16x32_f16 %[f], %[x], %[x], %[f]\n .endr\n .rept %c[nw]\n s_nop 0\n .endr\n"
#define BARRIER_ ".if %c[bar]\n s_barrier\n .endif\n"
#define OVER_LDS "ds_read_b128 %[r], %[addr]\n s_waitcnt lgkmcnt(0)\n s_nop 15\n s_nop 15\n"

template <int VICTIM, int DEP, int BAR, int NWAIT, int GAP, int PRIO>
__global__ __launch_bounds__(512) void war_kernel(int iters, unsigned *bad, int bcast) {
  __shared__ __attribute__((aligned(16))) float s_pat[8][64][4];   // pattern p: 16 bytes per lane
  __shared__ int s_done;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 8 * 64; i += 512) {
    const int p = i >> 6;
    if (VICTIM != 2) {
      half8 h;
      for (int j = 0; j < 8; ++j) h[j] = (_Float16)(float)(p + 1);
      *reinterpret_cast<half8 *>(s_pat[p][i & 63]) = h;
    } else {
      for (int j = 0; j < 4; ++j) s_pat[p][i & 63][j] = 1000.f * (p + 1);
    }
  }
  if (threadIdx.x == 0) s_done = 0;
  __syncthreads();
  half8 ones;
  for (int j = 0; j < 8; ++j) ones[j] = (_Float16)1.0f;
  if (wave >= 4) {
    if (PRIO == 1) __builtin_amdgcn_s_setprio(3);
    f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
    half8 w = ones;
    w[lane & 7] = (_Float16)0.5f;
    if (BAR) {
      // the decoder's situation at a slab boundary: all eight waves leave an s_barrier together and both waves of a
      // SIMD want the matrix pipe at once
      for (int it = 0; it < iters; ++it) {
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int i = 0; i < 24; ++i) {
          a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, ones, a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, ones, a1, 0, 0, 0);
        }
      }
    } else {
      do {
#pragma unroll
        for (int i = 0; i < 48; ++i) {
          a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, ones, a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, ones, a1, 0, 0, 0);
        }
      } while (*(volatile int *)&s_done < 4);
    }
    if (a0[0] + a1[0] == -1.f) bad[63] = 1;
  } else {
    unsigned nbad[6] = {0, 0, 0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
      const int p = it & 7, q = (it + 3) & 7;
      // LDS byte address of the NEW value; bcast: every lane reads the SAME 16 bytes (a conditioning-table read in the
      // decoder: one bank access, the shortest LDS return there is) instead of its own 16 of a 1-KiB fragment
      const unsigned addr = (unsigned)(size_t)(&s_pat[q][bcast ? 0 : lane][0]);
      f32x4 d0 = {0, 0, 0, 0}, d1 = d0, d2 = d0, d3 = d0, d4 = d0, d5 = d0, f = d0;
      float expect;
      if (VICTIM != 2) {                    // R = SrcA (0) or SrcB (1), f16 fragment
        half8 r = *reinterpret_cast<half8 *>(s_pat[p][lane]);
        const unsigned nv = 0x44004400u;    // 4.0, 4.0
        if (DEP == 0) {
          if (VICTIM == 0)
            asm volatile("s_nop 4\n" BARRIER_ INDEP("%[r]", "%[b]", "0") FILL OVER_LDS
                         : [d0] "+v"(d0), [d1] "+v"(d1), [d2] "=&v"(d2), [d3] "=&v"(d3), [d4] "=&v"(d4), [d5] "=&v"(d5),
                           [r] "+v"(r), [f] "+v"(f)
                         : [b] "v"(ones), [x] "v"(ones), [addr] "v"(addr), [nw] "n"(NWAIT), [gap] "n"(GAP), [bar] "n"(BAR), [nv] "v"(nv) : "memory");
          else
            asm volatile("s_nop 4\n" BARRIER_ INDEP("%[b]", "%[r]", "0") FILL OVER_LDS
                         : [d0] "+v"(d0), [d1] "+v"(d1), [d2] "=&v"(d2), [d3] "=&v"(d3), [d4] "=&v"(d4), [d5] "=&v"(d5),
                           [r] "+v"(r), [f] "+v"(f)
                         : [b] "v"(ones), [x] "v"(ones), [addr] "v"(addr), [nw] "n"(NWAIT), [gap] "n"(GAP), [bar] "n"(BAR), [nv] "v"(nv) : "memory");
          expect = 32.f * (p + 1);
        } else {
          if (VICTIM == 0)
            asm volatile("s_nop 4\n" BARRIER_ CHAIN("%[r]", "%[b]") FILL OVER_LDS
                         : [d0] "+v"(d0), [d1] "+v"(d1), [r] "+v"(r), [f] "+v"(f)
                         : [b] "v"(ones), [x] "v"(ones), [addr] "v"(addr), [nw] "n"(NWAIT), [gap] "n"(GAP), [bar] "n"(BAR), [nv] "v"(nv) : "memory");
          else
            asm volatile("s_nop 4\n" BARRIER_ CHAIN("%[b]", "%[r]") FILL OVER_LDS
                         : [d0] "+v"(d0), [d1] "+v"(d1), [r] "+v"(r), [f] "+v"(f)
                         : [b] "v"(ones), [x] "v"(ones), [addr] "v"(addr), [nw] "n"(NWAIT), [gap] "n"(GAP), [bar] "n"(BAR), [nv] "v"(nv) : "memory");
          expect = 3.f * 32.f * (p + 1);
        }
        if (r[0] != (_Float16)(float)(q + 1)) nbad[5] += 1u << 16;      // the overwrite itself must have happened
      } else {                              // R = SrcC (f32), independent MFMAs
        f32x4 r = *reinterpret_cast<f32x4 *>(s_pat[p][lane]);
        const float nv = 7777.f;
        asm volatile("s_nop 4\n" BARRIER_ INDEP("%[a]", "%[b]", "%[r]") FILL OVER_LDS
                     : [d0] "+v"(d0), [d1] "+v"(d1), [d2] "=&v"(d2), [d3] "=&v"(d3), [d4] "=&v"(d4), [d5] "=&v"(d5),
                       [r] "+v"(r), [f] "+v"(f)
                     : [a] "v"(ones), [b] "v"(ones), [x] "v"(ones), [addr] "v"(addr), [nw] "n"(NWAIT), [gap] "n"(GAP), [bar] "n"(BAR), [nv] "v"(nv) : "memory");
        expect = 32.f + 1000.f * (p + 1);
        if (r[0] != 1000.f * (q + 1)) nbad[5] += 1u << 16;
      }
      const f32x4 *d[6] = {&d0, &d1, &d2, &d3, &d4, &d5};
#pragma unroll
      for (int i = 0; i < (DEP ? 2 : 6); ++i) {
        const f32x4 v = *d[i];
        if (v[0] != expect || v[1] != expect || v[2] != expect || v[3] != expect) nbad[i]++;
      }
      if (f[0] != 32.f * GAP) nbad[4] += 1u << 20;                        // filler chain sanity
    }
    for (int i = 0; i < 6; ++i)
      if (nbad[i]) atomicAdd(&bad[i], nbad[i]);
    if (lane == 0) atomicAdd(&s_done, 1);
  }
}

static unsigned g_total = 0;
static int g_bcast = 0;

template <int VICTIM, int DEP, int BAR, int NWAIT, int GAP, int PRIO>
static void run(unsigned *d_bad, int iters) {
  (void)hipMemset(d_bad, 0, 64 * sizeof(unsigned));
  hipLaunchKernelGGL((war_kernel<VICTIM, DEP, BAR, NWAIT, GAP, PRIO>), dim3(512), dim3(512), 0, 0, iters, d_bad, g_bcast);
  unsigned h[64];
  if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); exit(1); }
  (void)hipMemcpy(h, d_bad, sizeof(h), hipMemcpyDeviceToHost);
  unsigned tot = 0;
  printf("%s R=%s %s %s gap=%d nwait=%2d prio=%s : bad lane-results per checked MFMA", g_bcast ? "bcast" : "frag ", VICTIM == 0 ? "SrcA" : VICTIM == 1 ? "SrcB" : "SrcC",
         DEP ? "chains" : "indep ", BAR ? "after-barrier" : "free-running ", GAP, NWAIT, PRIO == 0 ? "equal      " : "aggr-raised");
  for (int i = 0; i < 6; ++i) { printf(" %u", h[i]); tot += h[i]; }
  printf("  %s\n", tot ? "BAD" : "ok");
  g_total += tot;
}

template <int VICTIM, int DEP, int BAR, int GAP, int PRIO>
static void waits(unsigned *d_bad, int iters) {
  run<VICTIM, DEP, BAR, 0, GAP, PRIO>(d_bad, iters);
  run<VICTIM, DEP, BAR, 4, GAP, PRIO>(d_bad, iters);
  run<VICTIM, DEP, BAR, 16, GAP, PRIO>(d_bad, iters);
  run<VICTIM, DEP, BAR, 64, GAP, PRIO>(d_bad, iters);
}

template <int VICTIM, int DEP, int BAR, int PRIO>
static void gaps(unsigned *d_bad, int iters) {
  waits<VICTIM, DEP, BAR, 0, PRIO>(d_bad, iters);
  waits<VICTIM, DEP, BAR, 1, PRIO>(d_bad, iters);
  waits<VICTIM, DEP, BAR, 2, PRIO>(d_bad, iters);
  waits<VICTIM, DEP, BAR, 6, PRIO>(d_bad, iters);
}

int main(int argc, char **argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 3000;
  g_bcast = argc > 2 ? atoi(argv[2]) : 0;
  unsigned *d_bad;
  (void)hipMalloc(&d_bad, 64 * sizeof(unsigned));
  // LDS overwrite of SrcC / SrcA / SrcB, independent MFMAs
  waits<2, 0, 0, 0, 0>(d_bad, iters);
  waits<2, 0, 0, 0, 1>(d_bad, iters);
  waits<0, 0, 0, 0, 0>(d_bad, iters);
  waits<0, 0, 0, 0, 1>(d_bad, iters);
  waits<1, 0, 0, 0, 1>(d_bad, iters);
  // the decoder's shape: dependent chains, overwrite of SrcA / SrcB, with 0..6 filler MFMAs in between
  gaps<0, 1, 0, 0>(d_bad, iters);
  gaps<0, 1, 0, 1>(d_bad, iters);
  gaps<1, 1, 0, 0>(d_bad, iters);
  gaps<1, 1, 0, 1>(d_bad, iters);
  // ... and right behind a barrier release (where the round-2 failing build had its sunk MFMAs and the loads on top)
  gaps<0, 1, 1, 0>(d_bad, iters);
  gaps<0, 1, 1, 1>(d_bad, iters);
  gaps<1, 1, 1, 0>(d_bad, iters);
  gaps<1, 1, 1, 1>(d_bad, iters);
  waits<2, 0, 1, 0, 1>(d_bad, iters);
  printf("TOTAL bad %u (%d iterations x 512 workgroups x 4 victim waves per configuration)\n", g_total, iters);
  return 0;
}
