// boxes.hip -- proposal post-processing on the device: points-in-oriented-box
// counting and greedy class-aware 3-D NMS.
//
// Replaces the CPU numpy/scipy stage between detection and completion in the
// reference (net_utils/ap_helper.py:131-264 parse_predictions):
//   * remove_empty_box (:186-197): for each of the K boxes a scipy Delaunay
//     triangulation of its 8 corners + find_simplex over all N scan points
//     (net_utils/libs.py:128-137), keeping boxes with >= 5 points.  A box's
//     convex hull is the box itself, so this is a point-in-oriented-box test:
//     |(p-c).u| <= l/2, |(p-c).v| <= w/2, |p_z-c_z| <= h/2 with u = (cos a,
//     sin a, 0), v = (-sin a, cos a, 0) in the scan's (depth) frame.
//   * nms_3d_faster_samecls (net_utils/nms.py:79-118): boxes sorted by
//     objectness probability; the best survivor is picked, every other
//     survivor OF THE SAME CLASS whose axis-aligned IoU with it exceeds
//     nms_iou is dropped.
// K = 256 and N = 80 000: 20 M inside-tests (bandwidth of the L2-resident scan)
// and a K-step serial pick loop inside one workgroup.  Arithmetic in double
// like the reference's numpy.
#include "common.h"

namespace {

// one workgroup per box; threads stride over the points; xyz stride in floats
__global__ __launch_bounds__(256) void points_in_boxes_kernel(
    int n, int stride, const float *__restrict__ pts, const double *__restrict__ boxes /* K x 7 */,
    int *__restrict__ counts) {
  const int k = blockIdx.x, bi = blockIdx.y, K = gridDim.x;
  const double *b = boxes + ((size_t)bi * K + k) * 7;
  const double cx = b[0], cy = b[1], cz = b[2], hl = 0.5 * b[3], hw = 0.5 * b[4], hh = 0.5 * b[5];
  const double ca = cos(b[6]), sa = sin(b[6]);
  const float *p = pts + (size_t)bi * n * stride;
  int c = 0;
  for (int i = threadIdx.x; i < n; i += 256) {
    const double dx = (double)p[(size_t)i * stride + 0] - cx;
    const double dy = (double)p[(size_t)i * stride + 1] - cy;
    const double dz = (double)p[(size_t)i * stride + 2] - cz;
    const double u = dx * ca + dy * sa, v = -dx * sa + dy * ca;
    c += (fabs(u) <= hl && fabs(v) <= hw && fabs(dz) <= hh) ? 1 : 0;
  }
  __shared__ int s_c[4];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off);
  if ((threadIdx.x & 63) == 0) s_c[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) counts[(size_t)bi * K + k] = s_c[0] + s_c[1] + s_c[2] + s_c[3];
}

// aabb: (K,6) double [x1,y1,z1,x2,y2,z2]; order (K) int = box indices by
// DESCENDING score (the caller sorts; nms.py:91 argsort + picking from the end);
// cls (K) int; valid (K) u8 (boxes that take part); keep (K) u8 out.  One
// workgroup per scene, thread j owns box j; the pick loop walks `order`.
__global__ __launch_bounds__(1024) void nms3d_samecls_kernel(
    int K, double thr, int old_type, int use_cls, const double *__restrict__ aabb,
    const int *__restrict__ order, const int *__restrict__ cls,
    const unsigned char *__restrict__ valid, unsigned char *__restrict__ keep) {
  const int bi = blockIdx.x, j = threadIdx.x;
  aabb += (size_t)bi * K * 6; order += (size_t)bi * K; cls += (size_t)bi * K;
  valid += (size_t)bi * K; keep += (size_t)bi * K;
  __shared__ unsigned char s_alive[1024];
  __shared__ double s_box[1024 * 6];
  __shared__ int s_cls[1024];
  const bool in = j < K;
  double x1 = 0, y1 = 0, z1 = 0, x2 = 0, y2 = 0, z2 = 0;
  int cj = -1;
  if (in) {
    x1 = aabb[j * 6 + 0]; y1 = aabb[j * 6 + 1]; z1 = aabb[j * 6 + 2];
    x2 = aabb[j * 6 + 3]; y2 = aabb[j * 6 + 4]; z2 = aabb[j * 6 + 5];
    cj = cls[j];
    s_box[j * 6 + 0] = x1; s_box[j * 6 + 1] = y1; s_box[j * 6 + 2] = z1;
    s_box[j * 6 + 3] = x2; s_box[j * 6 + 4] = y2; s_box[j * 6 + 5] = z2;
    s_cls[j] = cj;
    s_alive[j] = valid[j] != 0;
    keep[j] = 0;
  }
  const double area = (x2 - x1) * (y2 - y1) * (z2 - z1);
  __syncthreads();
  for (int r = 0; r < K; ++r) {
    const int i = order[r];              // uniform
    if (s_alive[i]) {                    // uniform (read before anyone can clear it: barrier below)
      __syncthreads();
      if (j == i) { keep[j] = 1; s_alive[j] = 0; }
      else if (in && s_alive[j]) {
        const double px1 = s_box[i * 6 + 0], py1 = s_box[i * 6 + 1], pz1 = s_box[i * 6 + 2];
        const double px2 = s_box[i * 6 + 3], py2 = s_box[i * 6 + 4], pz2 = s_box[i * 6 + 5];
        const double parea = (px2 - px1) * (py2 - py1) * (pz2 - pz1);
        const double l = fmax(0.0, fmin(x2, px2) - fmax(x1, px1));
        const double w = fmax(0.0, fmin(y2, py2) - fmax(y1, py1));
        const double h = fmax(0.0, fmin(z2, pz2) - fmax(z1, pz1));
        const double inter = l * w * h;
        double o = old_type ? inter / area : inter / (parea + area - inter);
        if (use_cls && s_cls[i] != cj) o = 0.0;
        if (o > thr) s_alive[j] = 0;
      }
      __syncthreads();
    }
  }
}

}  // namespace

RFD_API int rfd_points_in_boxes(int b, int K, int n, int point_stride, const float *pts,
                                const double *boxes, int *counts, void *stream) {
  if (b <= 0 || K <= 0) return 0;
  hipLaunchKernelGGL(points_in_boxes_kernel, dim3(K, b), dim3(256), 0, (hipStream_t)stream, n,
                     point_stride, pts, boxes, counts);
  RFD_CHECK_LAUNCH();
  return 0;
}

RFD_API int rfd_nms3d(int b, int K, double iou_thr, int old_type, int use_cls, const double *aabb,
                      const int *order, const int *cls, const unsigned char *valid,
                      unsigned char *keep, void *stream) {
  if (b <= 0 || K <= 0) return 0;
  if (K > 1024) { rfd_set_error("rfd_nms3d: K > 1024", hipErrorInvalidValue); return (int)hipErrorInvalidValue; }
  const int threads = ((K + 63) / 64) * 64;
  hipLaunchKernelGGL(nms3d_samecls_kernel, dim3(b), dim3(threads), 0, (hipStream_t)stream, K, iou_thr,
                     old_type, use_cls, aabb, order, cls, valid, keep);
  RFD_CHECK_LAUNCH();
  return 0;
}
