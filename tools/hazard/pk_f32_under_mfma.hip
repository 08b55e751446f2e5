// pk_f32_under_mfma.hip -- packed-fp32 VALU instructions with op_sel beside another wave's matrix instructions (gfx950).
//   hipcc --offload-arch=gfx950 -O2 -w -o pk_f32_under_mfma pk_f32_under_mfma.hip
// Follow-up of load_valu_under_mfma.hip, which found v_pk_fma_f32 ... op_sel:[0,1,0] returning wrong values in LANES 48-63 while the
// SIMD's other wave runs v_mfma chains -- with the operands long landed (a gap of nops changes nothing).  Here no memory instruction is
// involved at all: operands are constants in registers.  A workgroup = eight waves, waves 0-3 run the VALU form under test `iters` times
// and accumulate, waves 4-7 (their SIMD partners) run dependent v_mfma_f32_16x16x32_f16 chains (or idle).  Every lane must end with the
// same exact sum; the table counts wrong lanes by quarter of the wave.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// FORM: what the reader executes per iteration; each adds exactly 2.0 to sum (w = (1, 1), p = (0, 1) or (1, 0) as the form needs)
enum { PKFMA_OPSEL_010 = 0, PKFMA_PLAIN, PKFMA_OPSELHI_101, PKMUL_OPSEL_01, PKADD_OPSEL_01, FMA_SCALAR, PKFMA_OPSEL_010_SINGLE, PKMOV_OPSEL_10,
       FMAMIX_OPSEL_100, PKFMA_F16_OPSEL_010, PKADD_U16_OPSEL_01, N_FORMS };
static const char *form_name[N_FORMS] = {
    "v_pk_fma_f32 op_sel:[0,1,0] x2 (dependent)", "v_pk_fma_f32 (no op_sel) x2            ", "v_pk_fma_f32 op_sel_hi:[1,0,1] x2       ",
    "v_pk_mul_f32 op_sel:[0,1] + v_pk_add    ", "v_pk_add_f32 op_sel:[0,1] x2            ", "v_fma_f32 x4 (scalar control)           ",
    "v_pk_fma_f32 op_sel:[0,1,0] x1, s_nop 4 ", "v_pk_mov_b32 op_sel:[1,0] + v_pk_add     ",
    // 16-bit forms the library DOES use beside matrix instructions (the decoder's and the GEMM's hi / lo split), and two more packed families
    "v_fma_mix_f32 op_sel:[1,0,0] x2         ", "v_pk_fma_f16 op_sel:[0,1,0] x2          ", "v_pk_add_u16 op_sel:[0,1] x2            "};

template <int FORM, int PARTNER>
__global__ __launch_bounds__(512) void probe(float *out, int iters, float one) {
  const int wave = threadIdx.x >> 6;
  float sum = 0.f;
  if (wave < 4) {
    f32x2 w = {one, one}, p01 = {0.f, one}, p10 = {one, 0.f}, acc = {0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
      asm volatile("" : "+v"(w), "+v"(p01), "+v"(p10));
      if (FORM == PKFMA_OPSEL_010) {
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]\n\tv_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc) : "v"(w), "v"(p01));
        acc *= 0.5f;
      } else if (FORM == PKFMA_PLAIN) {
        f32x2 p11 = {one, one};
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0\n\tv_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(p11));
        acc *= 0.5f;
      } else if (FORM == PKFMA_OPSELHI_101) {
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(w), "v"(p10));
        acc *= 0.5f;
      } else if (FORM == PKMUL_OPSEL_01) {
        f32x2 t;
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(t) : "v"(w), "v"(p01));
        acc += t;
      } else if (FORM == PKADD_OPSEL_01) {
        asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1]\n\tv_pk_add_f32 %0, %0, %1 op_sel:[0,1]" : "+v"(acc) : "v"(p01));
        acc *= 0.5f;
      } else if (FORM == FMA_SCALAR) {
        asm volatile("v_fma_f32 %0, %2, %3, %0\n\tv_fma_f32 %1, %2, %3, %1" : "+v"(acc[0]), "+v"(acc[1]) : "v"(w[0]), "v"(p01[1]));
      } else if (FORM == PKFMA_OPSEL_010_SINGLE) {
        asm volatile("s_nop 4\n\tv_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]\n\ts_nop 4" : "+v"(acc) : "v"(w), "v"(p01));
      } else if (FORM == PKMOV_OPSEL_10) {
        f32x2 t;
        asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(t) : "v"(p01), "v"(p10));      // (p01.hi, p10.lo) = (1, 1)
        acc += t;
      } else if (FORM == FMAMIX_OPSEL_100) {
        // r = float(h.hi) * 1 + 0 with h = (f16 0, f16 1) packed in one register: the high half selected by op_sel (the split's own form)
        unsigned h = 0x3c000000u;
        asm volatile("" : "+v"(h));
        float r0, r1;
        asm volatile("v_fma_mix_f32 %0, %2, 1.0, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %1, %2, 1.0, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                     : "=&v"(r0), "=&v"(r1) : "v"(h));
        acc[0] += r0;
        acc[1] += r1;
      } else if (FORM == PKFMA_F16_OPSEL_010) {
        // packed f16: acc.lo += w.lo * p.hi, acc.hi += w.hi * p.hi with w = (1, 1), p = (0, 1); two of them, then both halves to fp32
        unsigned w16 = 0x3c003c00u, p16 = 0x3c000000u, a16 = 0u;
        asm volatile("" : "+v"(w16), "+v"(p16));
        asm volatile("v_pk_fma_f16 %0, %1, %2, %0 op_sel:[0,1,0]\n\tv_pk_fma_f16 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(a16) : "v"(w16), "v"(p16));
        float lo, hi;
        asm volatile("s_nop 1\n\tv_cvt_f32_f16 %0, %2\n\tv_cvt_f32_f16_sdwa %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=&v"(lo), "=&v"(hi) : "v"(a16));
        acc[0] += 0.5f * lo;
        acc[1] += 0.5f * hi;
      } else {
        // packed u16 add: acc.lo += p.hi, acc.hi += p.hi with p = (0, 1); twice
        unsigned p16 = 0x00010000u, a16 = 0u;
        asm volatile("" : "+v"(p16));
        asm volatile("v_pk_add_u16 %0, %0, %1 op_sel:[0,1]\n\tv_pk_add_u16 %0, %0, %1 op_sel:[0,1]" : "+v"(a16) : "v"(p16));
        asm volatile("s_nop 1" : "+v"(a16));
        acc[0] += 0.5f * (float)(a16 & 0xffffu);
        acc[1] += 0.5f * (float)(a16 >> 16);
      }
      sum += acc[0] + acc[1];
      acc = f32x2{0.f, 0.f};
    }
  } else if (PARTNER) {
    half8 A, B;
    for (int i = 0; i < 8; ++i) A[i] = B[i] = (_Float16)one;
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
    for (int it = 0; it < iters; ++it)
      asm volatile("v_mfma_f32_16x16x32_f16 %0, %2, %3, %0\n\tv_mfma_f32_16x16x32_f16 %1, %2, %3, %1\n\t"
                   "v_mfma_f32_16x16x32_f16 %0, %2, %3, %0\n\tv_mfma_f32_16x16x32_f16 %1, %2, %3, %1\n\t"
                   "v_mfma_f32_16x16x32_f16 %0, %2, %3, %0\n\tv_mfma_f32_16x16x32_f16 %1, %2, %3, %1\n\t"
                   : "+v"(a0), "+v"(a1) : "v"(A), "v"(B));
    asm volatile("s_nop 15\n\ts_nop 15" : "+v"(a0), "+v"(a1));
    sum = a0[0] + a1[0];
  }
  out[(size_t)blockIdx.x * 512 + threadIdx.x] = sum;
}

template <int FORM, int PARTNER>
static void run(float *out, int n_wg, int iters) {
  (void)hipMemset(out, 0, (size_t)n_wg * 512 * sizeof(float));
  hipLaunchKernelGGL((probe<FORM, PARTNER>), dim3(n_wg), dim3(512), 0, 0, out, iters, 1.0f);
  (void)hipDeviceSynchronize();
  std::vector<float> h((size_t)n_wg * 512);
  (void)hipMemcpy(h.data(), out, h.size() * sizeof(float), hipMemcpyDeviceToHost);
  const float want = 2.f * iters;
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
  printf("%s partner %s: %5d waves, wrong %5ld, wrong lanes by quarter [%ld %ld %ld %ld], shortfall %.0f of %.3g\n", form_name[FORM],
         PARTNER ? "MFMA" : "idle", n_wg * 4, bad_waves, quarter[0], quarter[1], quarter[2], quarter[3], missing, (double)n_wg * 256 * want);
}

int main(int argc, char **argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 50000;
  float *out;
  const int n_wg = 256 * 2;
  (void)hipMalloc(&out, (size_t)n_wg * 512 * sizeof(float));
  run<PKFMA_OPSEL_010, 0>(out, n_wg, iters);
  run<PKFMA_OPSEL_010, 1>(out, n_wg, iters);
  run<PKFMA_PLAIN, 1>(out, n_wg, iters);
  run<PKFMA_OPSELHI_101, 1>(out, n_wg, iters);
  run<PKMUL_OPSEL_01, 1>(out, n_wg, iters);
  run<PKADD_OPSEL_01, 1>(out, n_wg, iters);
  run<FMA_SCALAR, 1>(out, n_wg, iters);
  run<PKFMA_OPSEL_010_SINGLE, 1>(out, n_wg, iters);
  run<PKMOV_OPSEL_10, 1>(out, n_wg, iters);
  run<FMAMIX_OPSEL_100, 1>(out, n_wg, iters);
  run<PKFMA_F16_OPSEL_010, 1>(out, n_wg, iters);
  run<PKADD_U16_OPSEL_01, 1>(out, n_wg, iters);
  return 0;
}
