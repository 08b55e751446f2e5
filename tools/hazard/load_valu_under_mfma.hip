// load_valu_under_mfma.hip -- can a VALU instruction issued right behind a satisfied s_waitcnt see a just-loaded register INCOMPLETE
// (the last lanes still holding the old value) while the SIMD's other wave keeps the matrix pipe busy?  Stand-alone probe:
//   hipcc --offload-arch=gfx950 -O2 -w -o load_valu_under_mfma load_valu_under_mfma.hip
// A workgroup has eight waves; waves 0-3 are READERS, waves 4-7 (the SIMD partners of 0-3) are MATRIX waves running back-to-back
// dependent v_mfma chains (or idling, as the control).  A reader repeats: zero four registers (v_mov), global_load_dwordx4 (or
// ds_read_b128) of four 1.0f into them from a line every wave reads (hot in L1), s_waitcnt vmcnt(0) / lgkmcnt(0), GAP x s_nop 0, then
// four v_add into a running sum.  Every lane must end with 4 * iters; a lane that saw a zero comes out short, and the per-lane table
// says which quarter of the wave it was.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int GAP, int LDS, int PARTNER, int CONSUMER>
__global__ __launch_bounds__(512) void probe(const float *ones, float *out, int iters) {
  __shared__ __attribute__((aligned(16))) float s_ones[64 * 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x < 256) s_ones[threadIdx.x] = 1.0f;
  __syncthreads();
  float sum = 0.f;
  if (wave < 4) {
    const float *src = ones + lane * 4;
    const unsigned lds_addr = (unsigned)(size_t)(s_ones + lane * 4);
    for (int it = 0; it < iters; ++it) {
      f32x4 v;
      if (LDS) {
        asm volatile("v_mov_b32 %0, 0\n\tv_mov_b32 %1, 0\n\tv_mov_b32 %2, 0\n\tv_mov_b32 %3, 0\n\t"
                     "s_nop 4\n\t"
                     "ds_read_b128 %4, %5\n\t"
                     "s_waitcnt lgkmcnt(0)\n\t"
                     : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v) : "v"(lds_addr) : "memory");
      } else {
        asm volatile("global_load_dwordx4 %0, %1, off\n\t"
                     "s_waitcnt vmcnt(0)\n\t"
                     : "=&v"(v) : "v"(src), "v"(0) : "memory");
      }
      if (GAP > 0) asm volatile("s_nop %0" ::"n"(GAP - 1));
      if (CONSUMER == 0) {
        asm volatile("v_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2\n\tv_add_f32 %0, %0, %3\n\tv_add_f32 %0, %0, %4"
                     : "+v"(sum) : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
      } else if (CONSUMER == 1) {
        // the shuffle hipcc builds (y0, y1) pairs with: D.lo = src0.hi, D.hi = src1.lo -- then the four sums as before
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x2 d0, d1;
        asm volatile("v_pk_mov_b32 %0, %2, %3 op_sel:[1,0]\n\tv_pk_mov_b32 %1, %3, %2 op_sel:[1,0]"
                     : "=&v"(d0), "=&v"(d1) : "v"(__builtin_shufflevector(v, v, 0, 1)), "v"(__builtin_shufflevector(v, v, 2, 3)));
        asm volatile("s_nop 1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2\n\tv_add_f32 %0, %0, %3\n\tv_add_f32 %0, %0, %4"
                     : "+v"(sum) : "v"(d0[0]), "v"(d0[1]), "v"(d1[0]), "v"(d1[1]));
      } else {
        // the y term: acc.lo += w.lo * p.hi, acc.hi += w.hi * p.hi  (op_sel:[0,1,0]) with p = (0, 1): adds w
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x2 acc = {0.f, 0.f}, pp = {0.f, 1.f};
        asm volatile("v_pk_fma_f32 %0, %1, %3, %0 op_sel:[0,1,0]\n\tv_pk_fma_f32 %0, %2, %3, %0 op_sel:[0,1,0]"
                     : "+v"(acc) : "v"(__builtin_shufflevector(v, v, 0, 1)), "v"(__builtin_shufflevector(v, v, 2, 3)), "v"(pp));
        asm volatile("s_nop 1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2" : "+v"(sum) : "v"(acc[0]), "v"(acc[1]));
      }
      // the destination holds zeros again before the next load (the load may be given the same registers)
      asm volatile("v_mov_b32 %0, 0\n\tv_mov_b32 %1, 0\n\tv_mov_b32 %2, 0\n\tv_mov_b32 %3, 0" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
      asm volatile("" ::"v"(v));
    }
  } else if (PARTNER) {
    half8 A, B;
    for (int i = 0; i < 8; ++i) A[i] = B[i] = (_Float16)1.0f;
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
    for (int it = 0; it < iters * 2; ++it) {
      asm volatile("v_mfma_f32_16x16x32_f16 %0, %2, %3, %0\n\tv_mfma_f32_16x16x32_f16 %1, %2, %3, %1\n\t"
                   "v_mfma_f32_16x16x32_f16 %0, %2, %3, %0\n\tv_mfma_f32_16x16x32_f16 %1, %2, %3, %1\n\t"
                   "v_mfma_f32_16x16x32_f16 %0, %2, %3, %0\n\tv_mfma_f32_16x16x32_f16 %1, %2, %3, %1\n\t"
                   : "+v"(a0), "+v"(a1) : "v"(A), "v"(B));
    }
    asm volatile("s_nop 15\n\ts_nop 15" : "+v"(a0), "+v"(a1));
    sum = a0[0] + a1[0];
  }
  out[(size_t)blockIdx.x * 512 + threadIdx.x] = sum;
}

template <int GAP, int LDS, int PARTNER, int CONSUMER>
static void run(const float *ones, float *out, int n_wg, int iters) {
  (void)hipMemset(out, 0, (size_t)n_wg * 512 * sizeof(float));
  hipLaunchKernelGGL((probe<GAP, LDS, PARTNER, CONSUMER>), dim3(n_wg), dim3(512), 0, 0, ones, out, iters);
  (void)hipDeviceSynchronize();
  std::vector<float> h((size_t)n_wg * 512);
  (void)hipMemcpy(h.data(), out, h.size() * sizeof(float), hipMemcpyDeviceToHost);
  const float want = 4.f * iters;
  long bad_waves = 0, quarter[4] = {0, 0, 0, 0};
  double missing = 0;
  for (int g = 0; g < n_wg; ++g)
    for (int w = 0; w < 4; ++w) {
      bool bad = false;
      for (int l = 0; l < 64; ++l) {
        const float v = h[(size_t)g * 512 + w * 64 + l];
        if (v != want) { bad = true; quarter[l >> 4]++; missing += want - v; }
      }
      bad_waves += bad;
    }
  printf("%s -> %s, gap %2d, partner %s: reader waves %6d, wrong %6ld, wrong lanes by quarter [%ld %ld %ld %ld], values missed %.0f of %.3g\n",
         LDS ? "ds_read_b128      " : "global_load_dwordx4", CONSUMER == 0 ? "v_add_f32            " : CONSUMER == 1 ? "v_pk_mov_b32 op_sel  " : "v_pk_fma_f32 op_sel  ", GAP, PARTNER ? "MFMA chains" : "idle       ", n_wg * 4, bad_waves,
         quarter[0], quarter[1], quarter[2], quarter[3], missing, (double)n_wg * 4 * 64 * want);
}

int main(int argc, char **argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  float *ones, *out;
  std::vector<float> h1(256, 1.0f);
  (void)hipMalloc(&ones, 256 * sizeof(float));
  (void)hipMemcpy(ones, h1.data(), 256 * sizeof(float), hipMemcpyHostToDevice);
  const int n_wg = 256 * 2;
  (void)hipMalloc(&out, (size_t)n_wg * 512 * sizeof(float));
  run<0, 0, 0, 0>(ones, out, n_wg, iters);
  run<0, 0, 1, 0>(ones, out, n_wg, iters);
  run<0, 0, 0, 1>(ones, out, n_wg, iters);
  run<0, 0, 1, 1>(ones, out, n_wg, iters);
  run<1, 0, 1, 1>(ones, out, n_wg, iters);
  run<4, 0, 1, 1>(ones, out, n_wg, iters);
  run<16, 0, 1, 1>(ones, out, n_wg, iters);
  run<0, 0, 0, 2>(ones, out, n_wg, iters);
  run<0, 0, 1, 2>(ones, out, n_wg, iters);
  run<4, 0, 1, 2>(ones, out, n_wg, iters);
  run<0, 1, 1, 0>(ones, out, n_wg, iters);
  run<0, 1, 0, 1>(ones, out, n_wg, iters);
  run<0, 1, 1, 1>(ones, out, n_wg, iters);
  run<4, 1, 1, 1>(ones, out, n_wg, iters);
  run<0, 1, 1, 2>(ones, out, n_wg, iters);
  return 0;
}
