// Second reproducer for the round-2 decoder failure (profiles/r03_decoder_hazard.txt section 7): the assembly-level bisect of the
// failing build ends in a window of the block-input code where a VALU instruction overwrites the first SrcA register of
// an MFMA issued two instructions earlier, 7 instructions behind the wave's previous MFMA:
//     v_mfma (own, keeps the matrix pipe busy) ; 6 x VALU ; v_mfma D, R, B, C ; VALU ; VALU writes R[0]
// Question: can that VALU write reach the register file before the second MFMA has read R -- when the SIMD's other wave
// streams MFMAs at a higher priority and the second MFMA has to wait for the matrix pipe?
//   victim waves 0-3:  PH wait states ; [PRE: an own MFMA + NV VALU ops] ; v_mfma d, r, ones, 0 ; G VALU ops ;
//                      v_mov r[REG] <- other value ; drain ; d must equal the product with the OLD r
//   aggressor waves 4-7: endless two-chain MFMA stream, s_setprio PRIO
// Build: hipcc --offload-arch=gfx950 -O2 -o rfdnet_amd/lib/micro/mfma_valu_war tools/micro/mfma_valu_war.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int PH, int PRE, int NV, int G, int REG, int PRIO>
__global__ __launch_bounds__(512) void k(int iters, unsigned *bad) {
  __shared__ int s_done;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) s_done = 0;
  __syncthreads();
  half8 ones;
  for (int j = 0; j < 8; ++j) ones[j] = (_Float16)1.0f;
  if (wave >= 4) {
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
    unsigned nbad = 0;
    for (int it = 0; it < iters; ++it) {
      const float pv = (float)((it & 7) + 1);
      union { _Float16 h[2]; unsigned u; } pat;
      pat.h[0] = pat.h[1] = (_Float16)pv;
      f32x4 d = {0, 0, 0, 0}, e = {0, 0, 0, 0};
      unsigned t0 = it, t1 = it + 1;
      const unsigned nv2 = 0x4c004c00u;    // 16.0, 16.0: not one of the patterns 1..8
      // R = v[20:23] by name, so that the overwrite can address ONE register of the MFMA's 128-bit operand
      asm volatile(
          "v_mov_b32 v20, %[pat]\n v_mov_b32 v21, %[pat]\n v_mov_b32 v22, %[pat]\n v_mov_b32 v23, %[pat]\n s_nop 7\n"
          ".rept %c[ph]\n s_nop 0\n .endr\n"
          ".if %c[pre]\n v_mfma_f32_16x16x32_f16 %[e], %[x], %[x], 0\n"
          " .rept %c[nvalu]\n v_add_u32 %[t0], %[t0], %[t1]\n .endr\n .endif\n"
          "v_mfma_f32_16x16x32_f16 %[d], v[20:23], %[x], 0\n"
          ".rept %c[g]\n v_add_u32 %[t1], %[t1], %[t0]\n .endr\n"
          ".if %c[reg] == 0\n v_mov_b32 v20, %[nv2]\n .else\n v_mov_b32 v23, %[nv2]\n .endif\n"
          "s_nop 15\n s_nop 15\n s_nop 15\n"
          : [d] "+v"(d), [e] "+v"(e), [t0] "+v"(t0), [t1] "+v"(t1)
          : [x] "v"(ones), [nv2] "v"(nv2), [pat] "v"(pat.u), [ph] "n"(PH), [pre] "n"(PRE), [nvalu] "n"(NV), [g] "n"(G), [reg] "n"(REG)
          : "memory", "v20", "v21", "v22", "v23");
      const float expect = 32.f * pv;
      if (d[0] != expect || d[1] != expect || d[2] != expect || d[3] != expect) nbad++;
      if (PRE && e[0] != 32.f) nbad += 1u << 16;
      if (t0 == 0xdeadbeef && t1 == 1) nbad += 1u << 24;
    }
    if (nbad) atomicAdd(&bad[0], nbad);
    if (lane == 0) atomicAdd(&s_done, 1);
  }
}

static unsigned g_total = 0;

template <int PH, int PRE, int NV, int G, int REG, int PRIO>
static void run(int iters, unsigned *bad) {
  hipMemset(bad, 0, 64 * sizeof(unsigned));
  hipLaunchKernelGGL((k<PH, PRE, NV, G, REG, PRIO>), dim3(512), dim3(512), 0, 0, iters, bad);
  unsigned h[64];
  hipMemcpy(h, bad, sizeof(h), hipMemcpyDeviceToHost);
  if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); exit(2); }
  g_total += h[0];
  printf("ph=%2d pre=%d nvalu=%d gap=%d reg=%d prio=%d : bad %u %s\n", PH, PRE, NV, G, REG, PRIO, h[0], h[0] ? "BAD" : "ok");
}

template <int PRE, int NV, int G, int REG, int PRIO>
static void phases(int iters, unsigned *bad) {
  run<0, PRE, NV, G, REG, PRIO>(iters, bad);
  run<1, PRE, NV, G, REG, PRIO>(iters, bad);
  run<2, PRE, NV, G, REG, PRIO>(iters, bad);
  run<3, PRE, NV, G, REG, PRIO>(iters, bad);
  run<5, PRE, NV, G, REG, PRIO>(iters, bad);
  run<7, PRE, NV, G, REG, PRIO>(iters, bad);
}

template <int PRE, int NV, int PRIO>
static void gaps(int iters, unsigned *bad) {
  phases<PRE, NV, 0, 0, PRIO>(iters, bad);
  phases<PRE, NV, 1, 0, PRIO>(iters, bad);
  phases<PRE, NV, 2, 0, PRIO>(iters, bad);
  phases<PRE, NV, 0, 3, PRIO>(iters, bad);
  phases<PRE, NV, 1, 3, PRIO>(iters, bad);
}

int main(int argc, char **argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;
  unsigned *bad;
  hipMalloc(&bad, 64 * sizeof(unsigned));
  gaps<0, 0, 0>(iters, bad);
  gaps<0, 0, 1>(iters, bad);
  gaps<0, 0, 3>(iters, bad);
  gaps<1, 6, 1>(iters, bad);      // the decoder's window: own MFMA, six VALU ops, the MFMA, one VALU, the overwrite
  gaps<1, 6, 3>(iters, bad);
  gaps<1, 4, 1>(iters, bad);
  gaps<1, 7, 1>(iters, bad);
  gaps<1, 6, 0>(iters, bad);
  printf("TOTAL bad %u (%d iterations x 512 workgroups x 4 victim waves per configuration)\n", g_total, iters);
  return 0;
}
