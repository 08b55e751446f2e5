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
    RFD_CHECK(hipMalloc((void **)&w->status, 64));
    RFD_CHECK(hipMemset(w->status, 0, 64));
    RFD_CHECK(hipMalloc((void **)&w->zeros, sizeof(float) * RFD_ZEROS_FLOATS));
    RFD_CHECK(hipMemset(w->zeros, 0, sizeof(float) * RFD_ZEROS_FLOATS));
    w->ring_pos.store(0);
    w->num_cu = 0;
    (void)hipDeviceGetAttribute(&w->num_cu, hipDeviceAttributeMultiprocessorCount, dev);
    g_ws[dev] = w;
  }
  *out = g_ws[dev];
  return 0;
}

RFD_API const char *rfd_last_error_string(void) { return g_last_error.c_str(); }

RFD_API int rfd_device_status(void) {
  RfdWorkspace *ws;
  if (rfd_get_workspace(&ws)) return -1;
  unsigned v = 0;
  if (hipDeviceSynchronize() != hipSuccess) return -2;
  if (hipMemcpy(&v, ws->status, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return -3;
  if (v) (void)hipMemset(ws->status, 0, sizeof(v));
  return (int)v;
}

// Same word, but waits only for `stream` (several scenes may be in flight on other streams).
RFD_API int rfd_stream_status(void *stream) {
  RfdWorkspace *ws;
  if (rfd_get_workspace(&ws)) return -1;
  unsigned v = 0;
  if (hipMemcpyAsync(&v, ws->status, sizeof(v), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess)
    return -3;
  if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return -2;
  if (v) (void)hipMemsetAsync(ws->status, 0, sizeof(v), (hipStream_t)stream);
  return (int)v;
}

RFD_API const char *rfd_build_arch(void) { return "gfx950"; }
