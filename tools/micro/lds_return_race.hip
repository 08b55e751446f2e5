// Fifth reproducer for the round-2 decoder failure (profiles/r03_decoder_hazard.txt section 9).  tools/fault_model.py
// explains ALL 60 wrong 16-point groups of a dump of the failing build by ONE parameter-free fault: in the tile prologue
// (fc_p: H' = row0 + Wp p) the y term of ONE channel is missing, always in lanes 48-63 (the last quarter of the wave),
// always a channel whose y weight goes through a register move (v_pk_mov_b32 / v_mov_b32) issued directly behind the
// `s_waitcnt lgkmcnt(n)` that covers the ds_read_b128 it reads -- while the SIMD's other wave, at a higher priority, is
// already in the block-input code (bursts of ds_read_b128 + MFMAs).  I.e. the move saw a stale register in the last
// 16 lanes although the wait count said the load was done.
//   victim waves 0-3:   ds_read_b128 R0 ; ds_read_b128 R1 ; ds_read2_b64 R2 ; s_waitcnt lgkmcnt(1) ;
//                       v_pk_mov_b32 X, R0[0:1], R1[0:1] op_sel:[1,0] ; v_mov_b32 Y, R1[3] ; ... compare with LDS contents;
//                       the registers are poisoned before every round, LDS contents change with the round
//   aggressor waves 4-7: s_setprio PRIO ; endless: 8 x ds_read_b128 burst, MFMAs on the loaded fragments, VALU conversions
// Build: hipcc --offload-arch=gfx950 -O2 -o rfdnet_amd/lib/micro/lds_return_race tools/micro/lds_return_race.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int PRIO, int NOPS, int AGG, int BC>
__global__ __launch_bounds__(512) void k(int iters, unsigned *bad) {
  __shared__ __attribute__((aligned(16))) unsigned s_pat[8][3][64][4];      // round, load, lane, dword
  __shared__ __attribute__((aligned(16))) half8 s_frag[64][64];             // aggressor's fragments (64 KiB)
  __shared__ int s_done;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 8 * 3 * 64 * 4; i += 512) (&s_pat[0][0][0][0])[i] = 0x10000000u + (unsigned)i * 2654435761u % 0x0fffffffu;
  for (int i = threadIdx.x; i < 64 * 64; i += 512) {
    half8 h;
    for (int j = 0; j < 8; ++j) h[j] = (_Float16)(0.001f * (float)((i + j) & 63));
    (&s_frag[0][0])[i] = h;
  }
  if (threadIdx.x == 0) s_done = 0;
  __syncthreads();
  if (wave >= 4) {
    if (!AGG) return;
    if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
    f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
    half8 b;
    for (int j = 0; j < 8; ++j) b[j] = (_Float16)1.0f;
    float t = 0.f;
    int r = wave;
    do {
#pragma unroll 1
      for (int rep = 0; rep < 16; ++rep) {
        half8 f[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = s_frag[(r + 8 * j + rep) & 63][lane];
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
          a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(f[j], b, a0, 0, 0, 0);
          t = __builtin_fmaf(t, 1.0001f, a1[0]);
          a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(f[j + 1], b, a1, 0, 0, 0);
          t = t > 1e30f ? 0.f : t;
        }
      }
      r += 3;
    } while (*(volatile int *)&s_done < 4);
    if (a0[0] + a1[0] + t == -1.f) bad[63] = 1;
  } else {
    unsigned nbad[4] = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
      const int rd = it & 7;
      // BC 1: the prologue's addressing -- the 16 lanes of a quarter read the SAME 16 bytes, the quarters 48 bytes apart,
      // the second load the adjacent 16 bytes (fc_p's weights of four channels per quarter)
      const int el = BC ? 3 * (lane >> 4) : lane;
      const unsigned a0 = (unsigned)(size_t)&s_pat[rd][0][el][0], a1 = (unsigned)(size_t)&s_pat[rd][0][el + 1][0],
                     a2 = (unsigned)(size_t)&s_pat[rd][BC ? 0 : 2][BC ? el + 2 : lane][0];
      unsigned x0, x1, y, z;
      const unsigned poison = 0x7fc00000u + (unsigned)it;
      asm volatile(
          "v_mov_b32 v20, %[po]\n v_mov_b32 v21, %[po]\n v_mov_b32 v22, %[po]\n v_mov_b32 v23, %[po]\n"
          "v_mov_b32 v24, %[po]\n v_mov_b32 v25, %[po]\n v_mov_b32 v26, %[po]\n v_mov_b32 v27, %[po]\n"
          "v_mov_b32 v28, %[po]\n v_mov_b32 v29, %[po]\n v_mov_b32 v30, %[po]\n v_mov_b32 v31, %[po]\n"
          "s_nop 4\n"
          "ds_read_b128 v[20:23], %[a0]\n"
          "ds_read_b128 v[24:27], %[a1]\n"
          "ds_read2_b64 v[28:31], %[a2] offset1:1\n"
          "s_waitcnt lgkmcnt(1)\n"
          ".rept %c[nops]\n s_nop 0\n .endr\n"
          "v_pk_mov_b32 v[32:33], v[20:21], v[24:25] op_sel:[1,0]\n"
          "v_mov_b32 v34, v27\n"
          "s_waitcnt lgkmcnt(0)\n"
          "v_mov_b32 v35, v31\n"
          "s_nop 2\n"
          "v_mov_b32 %[x0], v32\n v_mov_b32 %[x1], v33\n v_mov_b32 %[y], v34\n v_mov_b32 %[z], v35\n"
          : [x0] "=&v"(x0), [x1] "=&v"(x1), [y] "=&v"(y), [z] "=&v"(z)
          : [a0] "v"(a0), [a1] "v"(a1), [a2] "v"(a2), [po] "v"(poison), [nops] "n"(NOPS)
          : "memory", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33",
            "v34", "v35");
      if (x0 != s_pat[rd][0][el][1]) nbad[0]++;
      if (x1 != s_pat[rd][0][el + 1][0]) nbad[1]++;
      if (y != s_pat[rd][0][el + 1][3]) nbad[2]++;
      if (z != s_pat[rd][BC ? 0 : 2][BC ? el + 2 : lane][3]) nbad[3]++;
    }
    for (int i = 0; i < 4; ++i)
      if (nbad[i]) atomicAdd(&bad[4 * (lane >> 4) + i], nbad[i]);       // per quarter of the wave
    if (lane == 0) atomicAdd(&s_done, 1);
  }
}

static unsigned g_total = 0;

template <int PRIO, int NOPS, int AGG, int BC>
static void run(int iters, unsigned *bad) {
  (void)hipMemset(bad, 0, 64 * sizeof(unsigned));
  hipLaunchKernelGGL((k<PRIO, NOPS, AGG, BC>), dim3(256), dim3(512), 0, 0, iters, bad);
  unsigned h[64];
  (void)hipMemcpy(h, bad, sizeof(h), hipMemcpyDeviceToHost);
  if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); exit(2); }
  unsigned tot = 0;
  for (int i = 0; i < 16; ++i) tot += h[i];
  g_total += tot;
  printf("%s aggressor=%s wait_states_behind_waitcnt=%d : stale reads per lane quarter (pk_mov lo, pk_mov hi, mov, mov after full wait)",
         BC ? "quarter-broadcast addresses" : "lane-linear addresses      ", !AGG ? "none  " : PRIO == 0 ? "equal " : PRIO == 1 ? "prio 1" : "prio 3", NOPS);
  for (int q = 0; q < 4; ++q) printf("  q%d: %u %u %u %u", q, h[4 * q], h[4 * q + 1], h[4 * q + 2], h[4 * q + 3]);
  printf("  %s\n", tot ? "BAD" : "ok");
  fflush(stdout);
}

int main(int argc, char **argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  unsigned *bad;
  (void)hipMalloc(&bad, 64 * sizeof(unsigned));
  run<1, 0, 1, 1>(iters, bad);
  run<3, 0, 1, 1>(iters, bad);
  run<0, 0, 1, 1>(iters, bad);
  run<1, 1, 1, 1>(iters, bad);
  run<1, 0, 1, 0>(iters, bad);
  run<3, 0, 1, 0>(iters, bad);
  run<0, 0, 1, 0>(iters, bad);
  run<0, 0, 0, 0>(iters, bad);
  printf("TOTAL stale %u (%d rounds x 256 workgroups x 4 victim waves per configuration)\n", g_total, iters);
  return 0;
}
