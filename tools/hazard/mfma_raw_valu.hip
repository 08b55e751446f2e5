// mfma_raw_valu.hip -- is the result of a matrix instruction always there when a VALU instruction reads it after the wait states hipcc
// inserts (counted from the instruction's ISSUE), also when other waves share the SIMD's matrix pipe?  Stand-alone probe:
//   hipcc --offload-arch=gfx950 -O2 -w -o mfma_raw_valu mfma_raw_valu.hip
// Every wave: `iters` rounds of { acc = 0; CHAIN x acc = mfma(A, B, acc) (A = B = 1: +32 each); sum += acc } with compiler builtins, so
// the only thing between the last matrix instruction and the v_add that reads its result is what hipcc's hazard recogniser puts there.
// EXTRA > 0 adds that many s_nop 15 in front of the read.  sum must be 32 * CHAIN * iters in every element of every wave.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CHAIN, int EXTRA, int TWO>
__global__ __launch_bounds__(64) void probe(float *out, int iters, float one) {
  extern __shared__ char pad[];
  const int lane = threadIdx.x;
  half8 A, B;
  for (int i = 0; i < 8; ++i) A[i] = B[i] = (_Float16)one;
  f32x4 sum = {0.f, 0.f, 0.f, 0.f}, sum2 = {0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < CHAIN; ++c) {
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, acc, 0, 0, 0);
      if (TWO) acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, acc2, 0, 0, 0);
    }
    if (EXTRA >= 1) asm volatile("s_nop 15" : "+v"(acc), "+v"(acc2));
    if (EXTRA >= 2) asm volatile("s_nop 15" : "+v"(acc), "+v"(acc2));
    if (EXTRA >= 4) asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc), "+v"(acc2));
    if (EXTRA >= 8) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(acc), "+v"(acc2));
    sum += acc;
    if (TWO) sum2 += acc2;
    asm volatile("" : "+v"(A), "+v"(B));
  }
  float *o = out + ((size_t)blockIdx.x * 64 + lane) * 4;
  for (int i = 0; i < 4; ++i) o[i] = TWO ? 0.5f * (sum[i] + sum2[i]) : sum[i];
  if (pad == nullptr) o[0] = 0.f;
}

template <int CHAIN, int EXTRA, int TWO>
static void run(float *out, int n_wg, int iters, int lds) {
  (void)hipMemset(out, 0, (size_t)n_wg * 256 * sizeof(float));
  (void)hipFuncSetAttribute((const void *)probe<CHAIN, EXTRA, TWO>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL((probe<CHAIN, EXTRA, TWO>), dim3(n_wg), dim3(64), lds, 0, out, iters, 1.0f);
  (void)hipDeviceSynchronize();
  std::vector<float> h((size_t)n_wg * 256);
  (void)hipMemcpy(h.data(), out, h.size() * sizeof(float), hipMemcpyDeviceToHost);
  const float want = 32.f * CHAIN * iters;
  long bad_waves = 0;
  double off = 0;
  for (int w = 0; w < n_wg; ++w) {
    bool bad = false;
    for (int i = 0; i < 256; ++i)
      if (h[(size_t)w * 256 + i] != want) { bad = true; off += want - h[(size_t)w * 256 + i]; }
    bad_waves += bad;
  }
  printf("chain %d%s  extra s_nop 15 x %d  lds %6d (<= %2d waves/CU)  waves %5d  wrong waves %5ld  mean shortfall %.2f products per element\n",
         CHAIN, TWO ? " x2 accumulators" : "", EXTRA, lds, lds ? 163840 / lds : 32, n_wg, bad_waves,
         bad_waves ? off / (bad_waves * 256.0) / 32.0 : 0.0);
}

int main(int argc, char **argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 4000;
  float *out;
  const int n_wg = 256 * 32;
  (void)hipMalloc(&out, (size_t)n_wg * 256 * sizeof(float));
  const int ldss[] = {0, 20000, 40000, 80000, 160000};
  for (int lds : ldss) {
    run<1, 0, 0>(out, n_wg, iters, lds);
    run<3, 0, 0>(out, n_wg, iters, lds);
    run<3, 0, 1>(out, n_wg, iters, lds);
    run<6, 0, 1>(out, n_wg, iters, lds);
    run<3, 2, 1>(out, n_wg, iters, lds);
    run<3, 8, 1>(out, n_wg, iters, lds);
  }
  return 0;
}
