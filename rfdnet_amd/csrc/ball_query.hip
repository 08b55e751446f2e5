// ball_query.hip -- radius neighbour search for gfx950.
//
// Replaces _ext-src/src/ball_query_gpu.cu:9-44 (K4 query_ball_point_kernel).
// The reference gives each centre to ONE THREAD of a single block per scene and
// lets it stream all n points from global memory.  Here a 256-thread workgroup
// stages xyz tiles in LDS (SoA, conflict-free ds_read_b32) once for its
// CPW*4 centres; each WAVE owns CPW centres and tests 64 candidate points per
// step; `ballot` + prefix popcount hands out output slots in index order, so
// the result is exactly "the first nsample points k (ascending) with
// d2 < r^2, padded with the first hit; an all-zero row if there is none"
// (ball_query_gpu.cu:27-41, ball_query.cpp:19-21).  A workgroup stops
// streaming as soon as all of its centres are full.
#include "common.h"

namespace {

constexpr int BQ_THREADS = 256;
constexpr int BQ_TILE = 2048;  // points per LDS tile (24 KB)

template <int CPW>
__global__ __launch_bounds__(BQ_THREADS) void ball_query_kernel(
    int n, int m, float radius2, int nsample, const float *__restrict__ new_xyz,
    const float *__restrict__ xyz, int *__restrict__ idx) {
  __shared__ float s_x[BQ_TILE], s_y[BQ_TILE], s_z[BQ_TILE];
  const int bi = blockIdx.y;
  xyz += (size_t)bi * n * 3;
  new_xyz += (size_t)bi * m * 3;
  idx += (size_t)bi * m * nsample;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int j0 = (blockIdx.x * (BQ_THREADS / 64) + wave) * CPW;

  float cx[CPW], cy[CPW], cz[CPW];
  int cnt[CPW], first[CPW];
#pragma unroll
  for (int c = 0; c < CPW; ++c) {
    const int j = j0 + c;
    const bool in = j < m;
    cx[c] = in ? new_xyz[(size_t)j * 3 + 0] : 0.f;
    cy[c] = in ? new_xyz[(size_t)j * 3 + 1] : 0.f;
    cz[c] = in ? new_xyz[(size_t)j * 3 + 2] : 0.f;
    cnt[c] = in ? 0 : nsample;  // out-of-range centres are born full
    first[c] = 0;
  }
  const unsigned long long lt_mask = (1ull << lane) - 1ull;

  for (int tile0 = 0; tile0 < n; tile0 += BQ_TILE) {
    const int tn = (n - tile0) < BQ_TILE ? (n - tile0) : BQ_TILE;
    __syncthreads();  // previous tile fully consumed
    // coalesced flat copy of tn*3 floats, de-interleaved into SoA
    for (int f = t; f < tn * 3; f += BQ_THREADS) {
      const float v = xyz[(size_t)tile0 * 3 + f];
      const int p = f / 3, comp = f - p * 3;
      (comp == 0 ? s_x : comp == 1 ? s_y : s_z)[p] = v;
    }
    __syncthreads();
    bool wave_done = true;
#pragma unroll
    for (int c = 0; c < CPW; ++c) wave_done = wave_done && (cnt[c] >= nsample);
    if (!wave_done) {
      for (int p0 = 0; p0 < tn; p0 += 64) {
        const int p = p0 + lane;
        const bool in = p < tn;
        const float x = in ? s_x[p] : 0.f, y = in ? s_y[p] : 0.f, z = in ? s_z[p] : 0.f;
        const int k = tile0 + p;
        bool all_full = true;
#pragma unroll
        for (int c = 0; c < CPW; ++c) {
          if (cnt[c] < nsample) {  // wave-uniform
            const float d2 = sumsq3(cx[c] - x, cy[c] - y, cz[c] - z);  // :31-32
            const bool hit = in && (d2 < radius2);                     // :33
            const unsigned long long mk = __ballot(hit);
            if (mk) {
              if (cnt[c] == 0) first[c] = tile0 + p0 + (__ffsll((long long)mk) - 1);
              const int slot = cnt[c] + __popcll(mk & lt_mask);
              if (hit && slot < nsample) idx[(size_t)(j0 + c) * nsample + slot] = k;
              cnt[c] += __popcll(mk);
            }
            all_full = all_full && (cnt[c] >= nsample);
          }
        }
        if (all_full) break;
      }
    }
    bool done = true;
#pragma unroll
    for (int c = 0; c < CPW; ++c) done = done && (cnt[c] >= nsample);
    if (__syncthreads_and(done)) break;
  }
  // pad: first hit fills the unused slots; no hit => zeros (:34-38, host zero-init)
#pragma unroll
  for (int c = 0; c < CPW; ++c) {
    const int j = j0 + c;
    if (j < m && cnt[c] < nsample) {
      const int fill = cnt[c] == 0 ? 0 : first[c];
      for (int s = cnt[c] + lane; s < nsample; s += 64) idx[(size_t)j * nsample + s] = fill;
    }
  }
}

}  // namespace

RFD_API int query_ball_point_kernel_wrapper(int b, int n, int m, float radius,
                                            int nsample, const float *new_xyz,
                                            const float *xyz, int *idx,
                                            void *stream) {
  if (b <= 0 || m <= 0 || nsample <= 0) return 0;
  const float radius2 = radius * radius;  // ball_query_gpu.cu:22
  hipStream_t s = (hipStream_t)stream;
  // enough workgroups to cover the chip: 2 centres per wave when m is large
  if (m >= 2048) {
    hipLaunchKernelGGL(ball_query_kernel<2>, dim3(ceil_div(m, 8), b), dim3(BQ_THREADS), 0, s,
                       n, m, radius2, nsample, new_xyz, xyz, idx);
  } else {
    hipLaunchKernelGGL(ball_query_kernel<1>, dim3(ceil_div(m, 4), b), dim3(BQ_THREADS), 0, s,
                       n, m, radius2, nsample, new_xyz, xyz, idx);
  }
  RFD_CHECK_LAUNCH();
  return 0;
}
