// mfma_late_read.hip -- does a queued matrix instruction read its SrcA / SrcB registers AFTER a later-issued vector-memory load
// has written them?  (Stand-alone probe, hipcc --offload-arch=gfx950 -O2 -o mfma_late_read mfma_late_read.hip; not part of the library.)
//
// Every wave runs `iters` rounds of: a chain of CHAIN dependent v_mfma_f32_16x16x32_f16 (acc += A x B, A = B = 1.0: each adds 32 to
// every element), then IMMEDIATELY a global_load_dwordx4 of zeros into A's registers (the address is hot in L1: every wave reads the
// same 1 KiB), s_waitcnt vmcnt(0), a long s_nop tail (the chain has certainly drained), and A is restored from a copy.  If the matrix
// pipe read A when the instruction ISSUED, acc = 32 * CHAIN * iters exactly.  If a queued instruction reads A when it STARTS and the load's
// data can land first, some products use zeros and the sum comes out short.  Knobs: CHAIN, waves per CU (dynamic LDS), nops between the
// chain and the load, which operand the load hits (A = SrcA, B = SrcB, C = the accumulator of a second, independent chain).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MFMA "v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\t"
#define NOP16 "s_nop 15\n\t"

template <int CHAIN, int GAP, int TARGET>
__global__ __launch_bounds__(64) void probe(const half8 *zeros, float *out, int iters) {
  extern __shared__ char pad[];
  const int lane = threadIdx.x;
  half8 A, B, A0, B0;
  for (int i = 0; i < 8; ++i) A[i] = B[i] = (_Float16)1.0f;
  A0 = A;
  B0 = B;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const half8 *z = zeros + lane;
  for (int it = 0; it < iters; ++it) {
    if (TARGET == 0) {
      asm volatile(MFMA MFMA MFMA
                   : "+v"(acc) : "v"(A), "v"(B));
      if (CHAIN > 3) asm volatile(MFMA MFMA MFMA : "+v"(acc) : "v"(A), "v"(B));
      if (CHAIN > 6) asm volatile(MFMA MFMA MFMA MFMA MFMA MFMA : "+v"(acc) : "v"(A), "v"(B));
      if (GAP > 0) asm volatile("s_nop %0" ::"n"(GAP - 1));
      asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "+v"(A) : "v"(z) : "memory");
    } else if (TARGET == 1) {
      asm volatile(MFMA MFMA MFMA
                   : "+v"(acc) : "v"(A), "v"(B));
      if (CHAIN > 3) asm volatile(MFMA MFMA MFMA : "+v"(acc) : "v"(A), "v"(B));
      if (CHAIN > 6) asm volatile(MFMA MFMA MFMA MFMA MFMA MFMA : "+v"(acc) : "v"(A), "v"(B));
      if (GAP > 0) asm volatile("s_nop %0" ::"n"(GAP - 1));
      asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "+v"(B) : "v"(z) : "memory");
    } else {
      // SrcC: a chain in `tmp` (from zero), its LAST instruction writes `acc2 = A x B + tmp` into other registers, then the load
      // overwrites tmp.  acc2 must come out as 32 * per; a late SrcC read sees zeros: 32.
      f32x4 tmp = {0.f, 0.f, 0.f, 0.f}, acc2;
      asm volatile("s_nop 4" : "+v"(tmp));
      asm volatile(MFMA MFMA : "+v"(tmp) : "v"(A), "v"(B));
      if (CHAIN > 3) asm volatile(MFMA MFMA MFMA : "+v"(tmp) : "v"(A), "v"(B));
      if (CHAIN > 6) asm volatile(MFMA MFMA MFMA MFMA MFMA MFMA : "+v"(tmp) : "v"(A), "v"(B));
      asm volatile("v_mfma_f32_16x16x32_f16 %0, %2, %3, %1" : "=&v"(acc2), "+v"(tmp) : "v"(A), "v"(B));
      if (GAP > 0) asm volatile("s_nop %0" ::"n"(GAP - 1));
      asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "+v"(tmp) : "v"(z) : "memory");
      asm volatile(NOP16 NOP16 NOP16 NOP16 NOP16 NOP16 NOP16 NOP16 NOP16 NOP16 NOP16 NOP16 NOP16 NOP16 NOP16 NOP16 ::: "memory");
      acc += acc2;
    }
    asm volatile(NOP16 NOP16 NOP16 NOP16 NOP16 NOP16 NOP16 NOP16 NOP16 NOP16 NOP16 NOP16 NOP16 NOP16 NOP16 NOP16 ::: "memory");
    asm volatile("v_mov_b32 %0, %1" : "=v"(((unsigned *)&A)[0]) : "v"(((unsigned *)&A0)[0]));
    A = A0;
    B = B0;
    asm volatile("" : "+v"(A), "+v"(B));
    asm volatile(NOP16 ::: "memory");
  }
  asm volatile(NOP16 NOP16 NOP16 NOP16 ::: "memory");
  float *o = out + ((size_t)blockIdx.x * 64 + lane) * 4;
  o[0] = acc[0]; o[1] = acc[1]; o[2] = acc[2]; o[3] = acc[3];
  if (pad == nullptr) o[0] = 0.f;
}

template <int CHAIN, int GAP, int TARGET>
static void run(const half8 *zeros, float *out, int n_wg, int iters, int lds) {
  hipMemset(out, 0, (size_t)n_wg * 64 * 4 * sizeof(float));
  hipFuncSetAttribute((const void *)probe<CHAIN, GAP, TARGET>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL((probe<CHAIN, GAP, TARGET>), dim3(n_wg), dim3(64), lds, 0, zeros, out, iters);
  hipDeviceSynchronize();
  std::vector<float> h((size_t)n_wg * 64 * 4);
  hipMemcpy(h.data(), out, h.size() * sizeof(float), hipMemcpyDeviceToHost);
  const int per = CHAIN > 6 ? 12 : CHAIN > 3 ? 6 : 3;
  const float want = 32.f * per * iters;
  long bad_waves = 0;
  double short_sum = 0;
  for (int w = 0; w < n_wg; ++w) {
    bool bad = false;
    for (int i = 0; i < 256; ++i)
      if (h[(size_t)w * 256 + i] != want) { bad = true; short_sum += want - h[(size_t)w * 256 + i]; }
    bad_waves += bad;
  }
  printf("chain %2d  gap %2d  target %s  lds %6d (<= %2d waves/CU)  waves %5d  wrong waves %5ld  mean shortfall per wrong element %.1f products\n",
         per, GAP, TARGET == 0 ? "SrcA" : TARGET == 1 ? "SrcB" : "SrcC", lds, lds ? 163840 / lds : 32, n_wg, bad_waves,
         bad_waves ? short_sum / (bad_waves * 256.0) / 32.0 : 0.0);
}

int main(int argc, char **argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;
  half8 *zeros;
  float *out;
  hipMalloc(&zeros, 64 * sizeof(half8));
  hipMemset(zeros, 0, 64 * sizeof(half8));
  const int n_wg = 256 * 32;
  hipMalloc(&out, (size_t)n_wg * 64 * 4 * sizeof(float));
  const int ldss[] = {0, 20000, 80000, 160000};
  for (int lds : ldss) {
    run<3, 0, 0>(zeros, out, n_wg, iters, lds);
    run<6, 0, 0>(zeros, out, n_wg, iters, lds);
    run<12, 0, 0>(zeros, out, n_wg, iters, lds);
    run<6, 0, 1>(zeros, out, n_wg, iters, lds);
    run<3, 0, 2>(zeros, out, n_wg, iters, lds);
    run<6, 0, 2>(zeros, out, n_wg, iters, lds);
    run<12, 0, 2>(zeros, out, n_wg, iters, lds);
    run<6, 3, 2>(zeros, out, n_wg, iters, lds);
    run<6, 7, 2>(zeros, out, n_wg, iters, lds);
    run<6, 15, 2>(zeros, out, n_wg, iters, lds);
  }
  return 0;
}
