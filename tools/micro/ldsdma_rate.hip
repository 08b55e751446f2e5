// LDS-DMA fill rate from an L2-resident buffer: every wave of a 512-thread workgroup streams
// 1-KiB pieces (global_load_lds, 16 B per lane) of a small weight-like buffer into LDS.
// Build: hipcc --offload-arch=gfx950 -O3 -o ldsdma_rate tools/micro/ldsdma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((address_space(1))) void gbl_void;
typedef __attribute__((address_space(3))) void lds_void;

template <int INFLIGHT>
__global__ __launch_bounds__(512) void dma_kernel(const char *__restrict__ src, size_t bytes, int iters, int *sink) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[128 * 1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  size_t off = ((size_t)blockIdx.x * 7919 * 1024) % bytes;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < INFLIGHT; ++j) {
      const size_t o = (off + (size_t)(wave * INFLIGHT + j) * 1024) % bytes;
      __builtin_amdgcn_global_load_lds((gbl_void *)(src + o + lane * 16),
                                       (lds_void *)(smem + ((wave * INFLIGHT + j) & 127) * 1024), 16, 0, 0);
    }
    off = (off + 8 * INFLIGHT * 1024) % bytes;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  if (threadIdx.x == 0 && smem[blockIdx.x & 1023] == 77 && iters < 0) *sink = 1;
}

template <int INFLIGHT>
void run(const char *src, size_t bytes, int waves_label, int *sink) {
  const int iters = 2000, grid = 256;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(dma_kernel<INFLIGHT>, dim3(grid), dim3(512), 0, 0, src, bytes, 50, sink);
  hipEventRecord(a);
  hipLaunchKernelGGL(dma_kernel<INFLIGHT>, dim3(grid), dim3(512), 0, 0, src, bytes, iters, sink);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double total = (double)grid * 8 * INFLIGHT * 1024.0 * iters;
  printf("buffer %6.1f MB, %d pieces in flight per wave: %.2f TB/s chip, %.1f GB/s per CU\n", bytes / 1e6, INFLIGHT,
         total / ms / 1e9, total / ms / 1e6 / 256);
}

int main() {
  int *sink; hipMalloc(&sink, 4);
  for (size_t mb : {2, 8, 64}) {
    const size_t bytes = mb << 20;
    char *src; hipMalloc(&src, bytes + 65536); hipMemset(src, 1, bytes + 65536);
    run<1>(src, bytes, 8, sink);
    run<2>(src, bytes, 8, sink);
    run<4>(src, bytes, 8, sink);
    hipFree(src);
  }
  return 0;
}
