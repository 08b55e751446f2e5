// runtime.hip -- host-side plumbing of librfd_hip: error strings, per-device
// workspace, diagnostics.  No kernels here.
#include "common.h"

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
                        sizeof(unsigned long long) * (size_t)FPS_RING * FPS_REGION_GRANULES));
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
    g_ws[dev] = w;
  }
  *out = g_ws[dev];
  return 0;
}

RFD_API const char *rfd_last_error_string(void) { return g_last_error.c_str(); }

unsigned *rfd_status_word(RfdWorkspace *ws, hipStream_t stream) {
  void *key = (void *)stream;
  if (!key) return ws->status;                              // the null stream shares slot 0
  for (int i = 1; i < RFD_STATUS_SLOTS; ++i) {
    void *cur = ws->status_owner[i].load(std::memory_order_acquire);
    if (cur == key) return ws->status + i;
    if (!cur) {
      void *expected = nullptr;
      if (ws->status_owner[i].compare_exchange_strong(expected, key, std::memory_order_acq_rel)) return ws->status + i;
      if (expected == key) return ws->status + i;
    }
  }
  return ws->status;                                        // more than 63 streams: the shared word
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

RFD_API const char *rfd_build_arch(void) { return "gfx950"; }
