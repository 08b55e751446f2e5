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
// status words, claim counters).  Allocated once per device, never freed.
constexpr int RFD_ZEROS_FLOATS = 16384;
constexpr int RFD_STATUS_SLOTS = 64;
// FPS exchange regions.  A region = FPS_REGION_HEAD granules (granule 0 = the launch's sticky abort word, the
// rest padding to a 64-byte line of its own) + [workgroup][parity][5] exchange granules.  A stream that owns a
// status slot (1..63) owns region `slot`: launches on one stream are serial, so a region never has two users, however
// many multi-workgroup FPS launches are in flight on the device (round 4 dealt 16 regions round robin).  Streams
// on the shared slot 0 (the null stream; more than 63 streams) take turns on FPS_SHARED_RING regions behind them.
constexpr int RFD_FPS_TIMEOUT_MS_DEFAULT = 500;
constexpr int FPS_MAX_WG = 256;                         // co-resident WGs/launch
constexpr int FPS_REGION_HEAD = 8;
constexpr int FPS_REGION_GRANULES = FPS_REGION_HEAD + FPS_MAX_WG * 2 * 5;
constexpr int FPS_SHARED_RING = 16;
constexpr int FPS_REGIONS = RFD_STATUS_SLOTS + FPS_SHARED_RING;
// Counter pairs {next chunk, workgroups done} of the chunk-claiming kernels (occ_decoder8.hip): pair `slot` belongs
// to the stream owning status slot `slot` (serial launches: never shared); slot 0's users rotate
// over the rest of the pool.  Zero between launches (the last workgroup out resets its pair).
constexpr int RFD_CLAIM_SLOTS = 1024;
struct RfdWorkspace {
  unsigned long long *fps_slots;  // FPS_REGIONS regions of FPS_REGION_GRANULES
  unsigned *status;               // RFD_STATUS_SLOTS device status words (0 = OK): one per stream that has launched
                                  // a flag-raising kernel (rfd_status_slot), slot 0 = overflow / default
  std::atomic<void *> status_owner[RFD_STATUS_SLOTS];   // stream handle owning each slot (nullptr = free); slot 0 is shared
  float *zeros;                   // RFD_ZEROS_FLOATS zeros (stand-in for absent bias vectors)
  unsigned *claim;                // RFD_CLAIM_SLOTS counter pairs
  std::atomic<unsigned> claim_seq;  // round robin of the shared users (slot 0)
  std::atomic<unsigned> ring_pos;   // the same for the FPS regions
  int num_cu;                     // multiprocessor count of the device
  int wall_clock_khz;             // rate of wall_clock64() on the device (100 MHz on gfx950)
  std::atomic<int> fps_timeout_ms;  // multi-workgroup FPS: a workgroup that has polled this long for a round's
                                    // candidates aborts the launch (status bit 0); rfd_fps_set_timeout_ms
  std::atomic<int> fps_force_ppt;   // 0 = the launcher's own geometry; else points per thread (rfd_fps_set_geometry)
  std::atomic<int> fps_test_phantom;  // exchange units that never publish (rfd_fps_test_phantom_units; tests only)
};
int rfd_get_workspace(RfdWorkspace **ws);
// The status slot (0..RFD_STATUS_SLOTS-1) of `stream`: the slot it already owns, else a free one it claims now,
// else (null stream, table full) the shared slot 0.  Scenes in flight on different streams must not see (or
// clear) each other's flags: rfd_stream_status(stream) reads and resets this stream's word only.
int rfd_status_slot(RfdWorkspace *ws, hipStream_t stream);
static inline unsigned *rfd_status_word(RfdWorkspace *ws, hipStream_t stream) {
  return ws->status + rfd_status_slot(ws, stream);
}
static inline unsigned *rfd_claim_pair(RfdWorkspace *ws, hipStream_t stream) {
  const int slot = rfd_status_slot(ws, stream);
  if (slot > 0) return ws->claim + 2 * slot;
  return ws->claim + 2 * (RFD_STATUS_SLOTS + ws->claim_seq.fetch_add(1, std::memory_order_relaxed) %
                                                 (RFD_CLAIM_SLOTS - RFD_STATUS_SLOTS));
}
static inline unsigned long long *rfd_fps_region(RfdWorkspace *ws, hipStream_t stream) {
  const int slot = rfd_status_slot(ws, stream);
  const unsigned r = slot > 0 ? (unsigned)slot
                              : RFD_STATUS_SLOTS + ws->ring_pos.fetch_add(1, std::memory_order_relaxed) % FPS_SHARED_RING;
  return ws->fps_slots + (size_t)r * FPS_REGION_GRANULES;
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
