// runtime.hip -- host-side plumbing of librfd_hip: error strings, per-device
// workspace, diagnostics.  No kernels here.
#include "common.h"

#include <stdlib.h>

#include <mutex>
#include <string>

namespace {
thread_local std::string g_last_error;
std::mutex g_ws_mutex;
RfdWorkspace *g_ws[64] = {nullptr};
}  // namespace

void rfd_set_error(const char *where, hipError_t e) {
  g_last_error = std::string(where) + ": " + hipGetErrorString(e);
}

int rfd_get_workspace(RfdWorkspace **out) {
  int dev = 0;
  RFD_CHECK(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64) {
    rfd_set_error("rfd_get_workspace: device index", hipErrorInvalidDevice);
    return (int)hipErrorInvalidDevice;
  }
  std::lock_guard<std::mutex> lk(g_ws_mutex);
  if (!g_ws[dev]) {
    RfdWorkspace *w = new RfdWorkspace();
    RFD_CHECK(hipMalloc((void **)&w->fps_slots,
                        sizeof(unsigned long long) * (size_t)FPS_REGIONS * FPS_REGION_GRANULES));
    RFD_CHECK(hipMalloc((void **)&w->status, sizeof(unsigned) * RFD_STATUS_SLOTS));
    RFD_CHECK(hipMemset(w->status, 0, sizeof(unsigned) * RFD_STATUS_SLOTS));
    for (int i = 0; i < RFD_STATUS_SLOTS; ++i) w->status_owner[i].store(nullptr);
    RFD_CHECK(hipMalloc((void **)&w->zeros, sizeof(float) * RFD_ZEROS_FLOATS));
    RFD_CHECK(hipMemset(w->zeros, 0, sizeof(float) * RFD_ZEROS_FLOATS));
    RFD_CHECK(hipMalloc((void **)&w->claim, sizeof(unsigned) * 2 * RFD_CLAIM_SLOTS));
    RFD_CHECK(hipMemset(w->claim, 0, sizeof(unsigned) * 2 * RFD_CLAIM_SLOTS));
    w->claim_seq.store(0);
    w->ring_pos.store(0);
    w->num_cu = 0;
    (void)hipDeviceGetAttribute(&w->num_cu, hipDeviceAttributeMultiprocessorCount, dev);
    w->wall_clock_khz = 0;
    (void)hipDeviceGetAttribute(&w->wall_clock_khz, hipDeviceAttributeWallClockRate, dev);
    if (w->wall_clock_khz <= 0) w->wall_clock_khz = 100000;
    // RFD_FPS_TIMEOUT_MS: read once; rfd_fps_set_timeout_ms changes it at run time (tests)
    const char *to = getenv("RFD_FPS_TIMEOUT_MS");
    const int to_ms = to ? atoi(to) : 0;
    w->fps_timeout_ms.store(to_ms > 0 ? to_ms : RFD_FPS_TIMEOUT_MS_DEFAULT);
    w->fps_force_ppt.store(0);
    w->fps_test_phantom.store(0);
    g_ws[dev] = w;
  }
  *out = g_ws[dev];
  return 0;
}

RFD_API const char *rfd_last_error_string(void) { return g_last_error.c_str(); }

int rfd_status_slot(RfdWorkspace *ws, hipStream_t stream) {
  void *key = (void *)stream;
  if (!key) return 0;                                       // the null stream shares slot 0
  // Pass 1: the slot this stream already owns, wherever it is.  rfd_release_stream leaves free slots in the
  // MIDDLE of the table; claiming the first free slot before looking at the rest moved a live stream from slot 5
  // to a freed slot 3 -- flags raised in slot 5 by kernels still in flight were then never reported to it, and the
  // stream held two slots.
  for (int i = 1; i < RFD_STATUS_SLOTS; ++i)
    if (ws->status_owner[i].load(std::memory_order_acquire) == key) return i;
  // Pass 2: claim a free one.  (One stream is driven by one host thread at a time -- HIP's own rule for anything
  // that orders work on it -- so two threads do not race to claim two slots for the same stream.)
  for (int i = 1; i < RFD_STATUS_SLOTS; ++i) {
    void *expected = nullptr;
    if (ws->status_owner[i].load(std::memory_order_acquire) == nullptr &&
        ws->status_owner[i].compare_exchange_strong(expected, key, std::memory_order_acq_rel))
      return i;
  }
  return 0;                                                 // more than 63 streams: the shared word
}

// Every word (all streams): synchronises the DEVICE, returns the OR of the flags and clears them.
RFD_API int rfd_device_status(void) {
  RfdWorkspace *ws;
  if (rfd_get_workspace(&ws)) return -1;
  unsigned v[RFD_STATUS_SLOTS];
  if (hipDeviceSynchronize() != hipSuccess) return -2;
  if (hipMemcpy(v, ws->status, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return -3;
  unsigned all = 0;
  for (int i = 0; i < RFD_STATUS_SLOTS; ++i) all |= v[i];
  if (all) (void)hipMemset(ws->status, 0, sizeof(v));
  return (int)all;
}

// The word of `stream` only: waits for that stream (several scenes may be in flight on other streams, each with
// its own word, so a scene neither sees nor clears another scene's flags).  The shared word 0 belongs to the null
// stream (and to streams beyond the 63 slots): it is reported to THEM and to rfd_device_status only -- round 3 folded
// it read-only into every stream's answer, so one flag raised on the default stream made every scene in flight lower
// its scales or fail until somebody cleared it.
RFD_API int rfd_stream_status(void *stream) {
  RfdWorkspace *ws;
  if (rfd_get_workspace(&ws)) return -1;
  unsigned *word = rfd_status_word(ws, (hipStream_t)stream);
  unsigned v = 0;
  if (hipMemcpyAsync(&v, word, sizeof(unsigned), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) return -3;
  if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return -2;
  // stream-ordered reset: only kernels of this stream write this word (word 0: of the streams sharing it)
  if (v) (void)hipMemsetAsync(word, 0, sizeof(unsigned), (hipStream_t)stream);
  return (int)v;
}

// Asynchronous form for callers that must not wait here: the word of `stream` is copied to *host_word (pinned host
// memory, valid once the stream has been synchronised by whatever the caller waits for next) and reset, both in stream
// order -- flags raised by later launches are then distinguishable from the ones raised so far.
RFD_API int rfd_stream_status_snapshot(void *stream, unsigned *host_word) {
  RfdWorkspace *ws;
  if (rfd_get_workspace(&ws)) return -1;
  if (!host_word) return -3;
  unsigned *word = rfd_status_word(ws, (hipStream_t)stream);
  if (hipMemcpyAsync(host_word, word, sizeof(unsigned), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess)
    return -3;
  if (hipMemsetAsync(word, 0, sizeof(unsigned), (hipStream_t)stream) != hipSuccess) return -3;
  return 0;
}

// Give a stream's status slot back (a sweep that creates a stream per scene would otherwise run out of the 63 slots
// and fall back to the shared word).  Waits for the stream, returns its pending flags (like rfd_stream_status) and
// frees the slot; a stream that owns no slot is not an error.  The stream may be used again afterwards: it simply
// claims a slot again at its next flag-raising launch.
RFD_API int rfd_release_stream(void *stream) {
  RfdWorkspace *ws;
  if (rfd_get_workspace(&ws)) return -1;
  if (!stream) return 0;
  for (int i = 1; i < RFD_STATUS_SLOTS; ++i) {
    if (ws->status_owner[i].load(std::memory_order_acquire) != stream) continue;
    unsigned v = 0;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return -2;
    if (hipMemcpy(&v, ws->status + i, sizeof(unsigned), hipMemcpyDeviceToHost) != hipSuccess) return -3;
    if (v && hipMemset(ws->status + i, 0, sizeof(unsigned)) != hipSuccess) return -3;
    ws->status_owner[i].store(nullptr, std::memory_order_release);
    return (int)v;
  }
  return 0;
}

// ---- multi-workgroup FPS: time-out and geometry (diagnostics / tests) ------------------------------------------
// A launch whose G workgroups cannot all be resident (a CU-masked stream, a partitioned GPU, somebody else's
// persistent kernel holding the CUs) can never finish its first exchange; the reference's behaviour for a launch
// that cannot run is to fail fast (cuda_utils.h:30-39).  A workgroup that has polled `ms` for one round's
// candidates raises the launch's sticky abort word: every workgroup -- running or started later -- leaves, status
// bit 0 is set and the host raises.  Returns the previous value; ms <= 0 restores the default.
RFD_API int rfd_fps_set_timeout_ms(int ms) {
  RfdWorkspace *ws;
  if (rfd_get_workspace(&ws)) return -1;
  return ws->fps_timeout_ms.exchange(ms > 0 ? ms : RFD_FPS_TIMEOUT_MS_DEFAULT);
}

// Points per thread of the multi-workgroup kernel (0 = the launcher's choice).  Sweeps of the exchange geometry
// (tools/fps_sweep.py) and the time-out tests only; the result never depends on it.  Returns the previous value,
// -2 for a value that is not instantiated.
RFD_API int rfd_fps_set_geometry(int points_per_thread) {
  RfdWorkspace *ws;
  if (rfd_get_workspace(&ws)) return -1;
  switch (points_per_thread) {
    case 0: case 5: case 8: case 10: case 16: case 20: case 32: case 40: case 64: break;
    default: return -2;
  }
  return ws->fps_force_ppt.exchange(points_per_thread);
}

#ifndef RFD_NO_TEST_HOOKS
// TEST HOOK (one-shot: consumed by the next multi-workgroup FPS call, sampling.hip): every round of a multi-workgroup FPS launch also waits for `n` exchange units that nobody publishes -- the
// deterministic way to drive a launch into its exchange time-out (what a workgroup that is never dispatched looks like
// to the resident ones).  0 = off.  Returns the previous value.
RFD_API int rfd_fps_test_phantom_units(int n) {
  RfdWorkspace *ws;
  if (rfd_get_workspace(&ws)) return -1;
  return ws->fps_test_phantom.exchange(n < 0 ? 0 : n > 8 ? 8 : n);
}
#endif  // RFD_NO_TEST_HOOKS

RFD_API const char *rfd_build_arch(void) { return "gfx950"; }
