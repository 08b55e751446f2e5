// Third reproducer for the round-2 decoder failure (profiles/r03_decoder_hazard.txt section 7).  The assembly-level bisect of
// the failing build ends at an MFMA that reads, as SrcC, the result of an MFMA issued 15 instructions earlier INTO A
// DIFFERENT vDst (hipcc renamed the accumulator: v[96:99] = A x B + v[100:103]).  For that case ("XDL write VGPR ->
// XDL read SrcC, overlapped, not the same vDst") the hardware has no interlock: the compiler keeps a fixed number of
// wait states (passes + 2 .. 3) between the two and is done.  Question: does that fixed distance still hold when the
// SIMD's other wave streams MFMAs at a higher priority, i.e. can the producer be held up in front of the matrix pipe
// for longer than the wave needs to issue the wait states?
//   victim waves 0-3:  [PRE: an own MFMA + 3 VALU ops, so the pipe is busy when the producer arrives]
//                      producer  v_mfma p = x * x + p        (in place)
//                      NV VALU instructions                   (the wait states)
//                      consumer  v_mfma d = x * x + p        (different vDst, SrcC = the producer's vDst)
//                      d must be p_before + 64; p_before + 32 = the consumer read p before the producer wrote it
//   aggressor waves 4-7: endless two-chain MFMA stream at s_setprio PRIO
// NV below the documented distance must fail even without an aggressor (shows the check can see a stale read).
// Build: hipcc --offload-arch=gfx950 -O2 -o rfdnet_amd/lib/micro/mfma_srcc_raw tools/micro/mfma_srcc_raw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NV, int PRE, int PRIO, int AGG>
__global__ __launch_bounds__(512) void k(int iters, unsigned *bad) {
  __shared__ int s_done;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) s_done = 0;
  __syncthreads();
  half8 ones;
  for (int j = 0; j < 8; ++j) ones[j] = (_Float16)1.0f;
  if (wave >= 4) {
    if (!AGG) return;
    if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
    f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
    half8 w = ones;
    w[lane & 7] = (_Float16)0.5f;
    do {
#pragma unroll
      for (int i = 0; i < 48; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, ones, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, ones, a1, 0, 0, 0);
      }
    } while (*(volatile int *)&s_done < 4);
    if (a0[0] + a1[0] == -1.f) bad[63] = 1;
  } else {
    unsigned stale = 0, other = 0;
    for (int it = 0; it < iters; ++it) {
      const float c = (float)(it & 7);
      f32x4 p = {c, c, c, c}, d = {0, 0, 0, 0}, e = {0, 0, 0, 0};
      unsigned t0 = it, t1 = it + 1;
      asm volatile(
          "s_nop 7\n"
          ".if %c[pre]\n v_mfma_f32_16x16x32_f16 %[e], %[x], %[x], 0\n"
          " v_add_u32 %[t0], %[t0], %[t1]\n v_add_u32 %[t0], %[t0], %[t1]\n v_add_u32 %[t0], %[t0], %[t1]\n .endif\n"
          "v_mfma_f32_16x16x32_f16 %[p], %[x], %[x], %[p]\n"
          ".rept %c[nv]\n v_add_u32 %[t1], %[t1], %[t0]\n .endr\n"
          "v_mfma_f32_16x16x32_f16 %[d], %[x], %[x], %[p]\n"
          "s_nop 15\n s_nop 15\n s_nop 15\n"
          : [p] "+v"(p), [d] "=&v"(d), [e] "+v"(e), [t0] "+v"(t0), [t1] "+v"(t1)
          : [x] "v"(ones), [nv] "n"(NV), [pre] "n"(PRE)
          : "memory");
      if (d[0] == c + 32.f || d[3] == c + 32.f) stale++;
      else if (d[0] != c + 64.f || d[1] != c + 64.f || d[2] != c + 64.f || d[3] != c + 64.f) other++;
      if (p[0] != c + 32.f) other += 1u << 16;
      if (t0 == 0xdeadbeef && t1 == 1) other += 1u << 24;
    }
    if (stale) atomicAdd(&bad[0], stale);
    if (other) atomicAdd(&bad[1], other);
    if (lane == 0) atomicAdd(&s_done, 1);
  }
}

static unsigned g_total = 0;

template <int NV, int PRE, int PRIO, int AGG>
static void run(int iters, unsigned *bad) {
  (void)hipMemset(bad, 0, 64 * sizeof(unsigned));
  hipLaunchKernelGGL((k<NV, PRE, PRIO, AGG>), dim3(512), dim3(512), 0, 0, iters, bad);
  unsigned h[64];
  (void)hipMemcpy(h, bad, sizeof(h), hipMemcpyDeviceToHost);
  if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); exit(2); }
  if (NV >= 12) g_total += h[0] + h[1];
  printf("wait_states=%2d own_mfma_in_front=%d aggressor=%s : stale SrcC %u other %u %s\n", NV, PRE,
         !AGG ? "none      " : PRIO == 0 ? "equal prio" : PRIO == 1 ? "prio 1    " : "prio 3    ", h[0], h[1],
         (h[0] | h[1]) ? "BAD" : "ok");
  fflush(stdout);
}

template <int PRE, int PRIO, int AGG>
static void sweep(int iters, unsigned *bad) {
  run<2, PRE, PRIO, AGG>(iters, bad);
  run<4, PRE, PRIO, AGG>(iters, bad);
  run<6, PRE, PRIO, AGG>(iters, bad);
  run<8, PRE, PRIO, AGG>(iters, bad);
  run<10, PRE, PRIO, AGG>(iters, bad);
  run<12, PRE, PRIO, AGG>(iters, bad);
  run<15, PRE, PRIO, AGG>(iters, bad);
  run<20, PRE, PRIO, AGG>(iters, bad);
}

int main(int argc, char **argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 300;
  unsigned *bad;
  (void)hipMalloc(&bad, 64 * sizeof(unsigned));
  // the priority legs first (a starved victim is slow: keep iters small)
  sweep<1, 1, 1>(iters, bad);
  sweep<1, 3, 1>(iters, bad);
  sweep<0, 1, 1>(iters, bad);
  sweep<0, 3, 1>(iters, bad);
  sweep<1, 0, 1>(iters, bad);
  sweep<1, 0, 0>(iters, bad);
  printf("TOTAL bad at >= 12 wait states: %u (%d iterations x 512 workgroups x 4 victim waves per configuration)\n", g_total, iters);
  return 0;
}
