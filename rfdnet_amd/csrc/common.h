// common.h -- shared device/host helpers of librfd_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <atomic>

#define RFD_API extern "C" __attribute__((visibility("default")))

// ---- error plumbing ---------------------------------------------------------
void rfd_set_error(const char *where, hipError_t e);

#define RFD_CHECK(expr)                        \
  do {                                         \
    hipError_t _e = (expr);                    \
    if (_e != hipSuccess) {                    \
      rfd_set_error(#expr, _e);                \
      return (int)_e;                          \
    }                                          \
  } while (0)

#define RFD_CHECK_LAUNCH()                     \
  do {                                         \
    hipError_t _e = hipGetLastError();         \
    if (_e != hipSuccess) {                    \
      rfd_set_error(__func__, _e);             \
      return (int)_e;                          \
    }                                          \
  } while (0)

// Per-device scratch shared by the persistent kernels (FPS granule exchange,
// status word).  Allocated once per device, never freed.
struct RfdWorkspace {
  unsigned long long *fps_slots;  // FPS_RING regions of FPS_REGION_GRANULES
  unsigned *status;               // RFD_STATUS_SLOTS device status words (0 = OK): one per stream that has launched
                                  // a flag-raising kernel (rfd_status_word), slot 0 = overflow / default
  std::atomic<void *> status_owner[64];   // stream handle owning each slot (nullptr = free); slot 0 is shared
  float *zeros;                   // RFD_ZEROS_FLOATS zeros (stand-in for absent bias vectors)
  unsigned *claim;                // RFD_CLAIM_SLOTS pairs {next chunk, workgroups done} of the persistent kernels that
                                  // hand their tiles out dynamically (occ_decoder8.hip); zero between launches
  std::atomic<unsigned> claim_seq;
  std::atomic<unsigned> ring_pos;   // callers may come from several host threads / streams
  int num_cu;                     // multiprocessor count of the device
};
constexpr int RFD_ZEROS_FLOATS = 16384;
constexpr int FPS_RING = 16;
constexpr int FPS_MAX_WG = 256;                         // co-resident WGs/launch
constexpr int FPS_REGION_GRANULES = FPS_MAX_WG * 2 * 5; // [wg][parity][field]
constexpr int RFD_STATUS_SLOTS = 64;
constexpr int RFD_CLAIM_SLOTS = 1024;
int rfd_get_workspace(RfdWorkspace **ws);
// The status word kernels launched on `stream` raise their flags in.  Scenes in flight on different streams
// must not see (or clear) each other's flags: rfd_stream_status(stream) reads and resets this word only.
unsigned *rfd_status_word(RfdWorkspace *ws, hipStream_t stream);
// A zeroed counter pair for one launch of a chunk-claiming kernel (the kernel leaves it zeroed again).  Slots are
// dealt round robin, so launches in flight together never share one unless RFD_CLAIM_SLOTS of them overlap.
static inline unsigned *rfd_claim_pair(RfdWorkspace *ws) {
  return ws->claim + 2 * (ws->claim_seq.fetch_add(1, std::memory_order_relaxed) % RFD_CLAIM_SLOTS);
}

// ---- arithmetic contract ------------------------------------------------------
// a*a + b*b + c*c as nvcc -fmad=true contracts it (see oracle/rfd_oracle.c
// header): t = b*b; t = fma(a,a,t); t = fma(c,c,t).  The library is compiled
// with -ffp-contract=off so this order is exact.
__device__ __forceinline__ float sumsq3(float a, float b, float c) {
  float t = b * b;
  t = __builtin_fmaf(a, a, t);
  t = __builtin_fmaf(c, c, t);
  return t;
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// cuda_utils.h:13-19 -- needed host-side only to reproduce the FPS tie order.
static inline int ref_opt_n_threads(int work_size) {
  const int pow_2 = (int)(log((double)work_size) / log(2.0));
  int v = 1 << pow_2;
  if (v > 512) v = 512;
  if (v < 1) v = 1;
  return v;
}
