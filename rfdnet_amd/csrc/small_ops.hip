// small_ops.hip -- three elementwise passes that the host composed out of eight to ten torch launches each (gfx950).
// Not hot in bytes or flops: each replaces a run of ~4.5-us launches on the per-scene critical path
// (tools/launch_sequence.py, profiles/r06_single_scene_sequence.txt).
#include "common.h"

namespace {

typedef float so4 __attribute__((ext_vector_type(4)));

// ---- DecoderCBatchNorm's per-proposal table from the stacked gamma / beta products (rfdnet_amd/occ_fold.py
// fold_table_stacked; layers.py:226-242 CBN = gamma(c) * BN_eval(x) + beta(c), folded to scale / shift rows):
//   scale = gb[k][l] / sqrtv[l];  shift = gb[k][L + l] - mean[l] * scale
//   table[k][1 + 2l] = scale * smul[l];   table[k][2 + 2l] = (shift + scale * extra[l]) * tmul[l];   table[k][0] = row0[k]
// every operation rounded on its own, in the order the torch expressions evaluated them (bit-identical table).
__global__ __launch_bounds__(256) void occ_fold_rows_kernel(
    int K, int L, int H, const float *__restrict__ gb, const float *__restrict__ sqrtv, const float *__restrict__ mean,
    const float *__restrict__ extra, const float *__restrict__ smul, const float *__restrict__ tmul,
    const float *__restrict__ row0, int row0_stride, float *__restrict__ table) {
  const int h4 = H / 4;
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (size_t)K * (L + 1) * h4) return;
  const int c = (int)(e % h4) * 4;
  const int l = (int)((e / h4) % (L + 1)) - 1;          // -1: row 0
  const int k = (int)(e / ((size_t)h4 * (L + 1)));
  float *trow = table + (size_t)k * (2 * L + 1) * H;
  if (l < 0) {
    *reinterpret_cast<so4 *>(trow + c) = *reinterpret_cast<const so4 *>(row0 + (size_t)k * row0_stride + c);
    return;
  }
  const so4 g = *reinterpret_cast<const so4 *>(gb + ((size_t)k * 2 * L + l) * H + c);
  const so4 b = *reinterpret_cast<const so4 *>(gb + ((size_t)k * 2 * L + L + l) * H + c);
  const so4 sq = *reinterpret_cast<const so4 *>(sqrtv + (size_t)l * H + c);
  const so4 mn = *reinterpret_cast<const so4 *>(mean + (size_t)l * H + c);
  const so4 ex = *reinterpret_cast<const so4 *>(extra + (size_t)l * H + c);
  const float sm = smul[l], tm = tmul[l];
  so4 t1, t2;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float scale = g[i] / sq[i];
    const float ms = mn[i] * scale;
    const float shift = b[i] - ms;
    t1[i] = scale * sm;
    const float se = scale * ex[i];
    t2[i] = (shift + se) * tm;
  }
  *reinterpret_cast<so4 *>(trow + (size_t)(1 + 2 * l) * H + c) = t1;
  *reinterpret_cast<so4 *>(trow + (size_t)(2 + 2 * l) * H + c) = t2;
}

// ---- STN_Group (pointnet2_modules.py:517-527): rotate every group's points by -heading about z, then the learned
// 3 x 4 affine (:452-466).  rows [G][P][3]; mode 0: out = rows . R(angle)^T with R^T = [[c, -s, 0], [s, c, 0], [0, 0, 1]]
// (x' = x c + y s, y' = -x s + y c); mode 1: out = rows . A[:, :3]^T + A[:, 3] with A [G][3][4].
__global__ __launch_bounds__(256) void rows3_transform_kernel(int mode, int G, int P, const float *__restrict__ rows,
                                                             const float *__restrict__ par, float *__restrict__ out) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (size_t)G * P) return;
  const int g = (int)(e / P);
  const float x = rows[e * 3], y = rows[e * 3 + 1], z = rows[e * 3 + 2];
  float ox, oy, oz;
  if (mode == 0) {
    const float c = par[2 * g], s = par[2 * g + 1];           // (cos, sin) computed by the caller
    ox = __builtin_fmaf(y, s, x * c);
    oy = __builtin_fmaf(y, c, x * -s);
    oz = z;
  } else {
    const float *A = par + (size_t)g * 12;
    ox = __builtin_fmaf(z, A[2], __builtin_fmaf(y, A[1], x * A[0])) + A[3];
    oy = __builtin_fmaf(z, A[6], __builtin_fmaf(y, A[5], x * A[4])) + A[7];
    oz = __builtin_fmaf(z, A[10], __builtin_fmaf(y, A[9], x * A[8])) + A[11];
  }
  out[e * 3] = ox;
  out[e * 3 + 1] = oy;
  out[e * 3 + 2] = oz;
}

}  // namespace

// gb [K][2L][H] (gamma rows then beta rows), sqrtv / mean / extra [L][H], smul / tmul [L], row0 [K or 1][H]
// (row0_stride = H or 0), table [K][2L + 1][H]; H % 4 == 0, 16-byte aligned tensors.
RFD_API int rfd_occ_fold_rows(int K, int L, int H, const float *gb, const float *sqrtv, const float *mean,
                              const float *extra, const float *smul, const float *tmul, const float *row0,
                              int row0_stride, float *table, void *stream) {
  if (K <= 0 || L <= 0) return 0;
  if (H <= 0 || (H & 3) || ((uintptr_t)gb & 15) || ((uintptr_t)table & 15) || ((uintptr_t)row0 & 15) ||
      ((uintptr_t)sqrtv & 15) || ((uintptr_t)mean & 15) || ((uintptr_t)extra & 15) || (row0_stride & 3)) {
    rfd_set_error("rfd_occ_fold_rows: H % 4 == 0 and 16-byte aligned tensors", hipErrorInvalidValue);
    return (int)hipErrorInvalidValue;
  }
  const size_t n = (size_t)K * (L + 1) * (H / 4);
  hipLaunchKernelGGL(occ_fold_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, K, L, H,
                     gb, sqrtv, mean, extra, smul, tmul, row0, row0_stride, table);
  RFD_CHECK_LAUNCH();
  return 0;
}

// rows / out [G][P][3] fp32 (out may be rows); cs [G][2] = (cos, sin) of each group's angle.
RFD_API int rfd_rows3_rotate_z(int G, int P, const float *rows, const float *cs, float *out, void *stream) {
  if (G <= 0 || P <= 0) return 0;
  const size_t n = (size_t)G * P;
  hipLaunchKernelGGL(rows3_transform_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, 0, G, P,
                     rows, cs, out);
  RFD_CHECK_LAUNCH();
  return 0;
}

// out = rows . A[:, :3]^T + A[:, 3], A [G][3][4].
RFD_API int rfd_rows3_affine(int G, int P, const float *rows, const float *A, float *out, void *stream) {
  if (G <= 0 || P <= 0) return 0;
  const size_t n = (size_t)G * P;
  hipLaunchKernelGGL(rows3_transform_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, 1, G, P,
                     rows, A, out);
  RFD_CHECK_LAUNCH();
  return 0;
}
