// ball_query.hip -- radius neighbour search for gfx950.
//
// Replaces _ext-src/src/ball_query_gpu.cu:9-44 (K4 query_ball_point_kernel).
// The reference gives each centre to ONE THREAD of a single block per scene and
// lets it stream all n points from global memory.  Here a 256-thread workgroup
// owns CPW centres and stages xyz tiles in LDS (plain 16-byte copies);
// its four waves each test a QUARTER of the tile against all CPW centres (eight
// independent 64-point steps: `ballot` masks kept in scalar registers), publish
// their hit counts, and after one barrier hand out output slots in index order
// (wave order, then step order, then prefix popcount).  The result is exactly
// "the first nsample points k (ascending) with d2 < r^2, padded with the first
// hit; an all-zero row if there is none" (ball_query_gpu.cu:27-41,
// ball_query.cpp:19-21).  A workgroup stops streaming as soon as all of its
// centres are full.  (The first version gave every wave its own centres and a
// serial chain of 64-point steps with the slot counter in the loop: 0.4 ms for
// the two 80 000-point queries of a scene, latency bound at one wave per SIMD.)
#include "common.h"

namespace {

constexpr int BQ_THREADS = 256;
constexpr int BQ_WAVES = BQ_THREADS / 64;
constexpr int BQ_TILE = 2048;                       // points per LDS tile (24 KB)
constexpr int BQ_STEPS = BQ_TILE / BQ_THREADS;      // 64-point steps per wave and tile

template <int CPW>
__global__ __launch_bounds__(BQ_THREADS) void ball_query_kernel(
    int n, int m, float radius2, int nsample, const float *__restrict__ new_xyz,
    const float *__restrict__ xyz, int *__restrict__ idx) {
  __shared__ __attribute__((aligned(16))) float s_p[BQ_TILE * 3];   // xyz triples as in memory
  __shared__ int s_cnt[2][BQ_WAVES][CPW];   // hits of each wave in the current tile (double-buffered)
  __shared__ int s_first[CPW];              // index of a centre's very first hit
  const int bi = blockIdx.y;
  xyz += (size_t)bi * n * 3;
  new_xyz += (size_t)bi * m * 3;
  idx += (size_t)bi * m * nsample;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int j0 = blockIdx.x * CPW;

  float cx[CPW], cy[CPW], cz[CPW];
  int cnt[CPW];                             // hits so far (uniform over the workgroup)
#pragma unroll
  for (int c = 0; c < CPW; ++c) {
    const int j = j0 + c;
    const bool in = j < m;
    cx[c] = in ? new_xyz[(size_t)j * 3 + 0] : 0.f;
    cy[c] = in ? new_xyz[(size_t)j * 3 + 1] : 0.f;
    cz[c] = in ? new_xyz[(size_t)j * 3 + 2] : 0.f;
    cnt[c] = in ? 0 : nsample;  // out-of-range centres are born full
  }
  if (t < CPW) s_first[t] = 0;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;

  int par = 0;
  for (int tile0 = 0; tile0 < n; tile0 += BQ_TILE, par ^= 1) {
    const int tn = (n - tile0) < BQ_TILE ? (n - tile0) : BQ_TILE;
    __syncthreads();  // previous tile fully consumed
    // flat copy of tn*3 floats (16 B per thread and pass when the tile is aligned); a lane then
    // reads its point's x, y, z at a stride of three words -- odd, so conflict-free
    {
      const float *src = xyz + (size_t)tile0 * 3;
      const int nf = tn * 3;
      if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {
        const int nv = nf >> 2;
        for (int f = t; f < nv; f += BQ_THREADS)
          reinterpret_cast<float4 *>(s_p)[f] = reinterpret_cast<const float4 *>(src)[f];
        for (int f = (nv << 2) + t; f < nf; f += BQ_THREADS) s_p[f] = src[f];
      } else {
        for (int f = t; f < nf; f += BQ_THREADS) s_p[f] = src[f];
      }
    }
    __syncthreads();
    // ---- phase 1: hit masks of this wave's quarter of the tile
    unsigned long long mk[CPW][BQ_STEPS];
    int mine[CPW];
#pragma unroll
    for (int c = 0; c < CPW; ++c) mine[c] = 0;
    const int q0 = wave * (BQ_TILE / BQ_WAVES);
#pragma unroll
    for (int s = 0; s < BQ_STEPS; ++s) {
      const int p = q0 + s * 64 + lane;
      const bool in = p < tn;
      const float x = in ? s_p[3 * p] : 0.f, y = in ? s_p[3 * p + 1] : 0.f, z = in ? s_p[3 * p + 2] : 0.f;
#pragma unroll
      for (int c = 0; c < CPW; ++c) {
        const float d2 = sumsq3(cx[c] - x, cy[c] - y, cz[c] - z);  // :31-32
        mk[c][s] = __ballot(in && (d2 < radius2) && cnt[c] < nsample);   // :33
        mine[c] += __popcll(mk[c][s]);
      }
    }
    if (lane == 0) {
#pragma unroll
      for (int c = 0; c < CPW; ++c) s_cnt[par][wave][c] = mine[c];
    }
    __syncthreads();
    // ---- phase 2: slots in index order = (earlier tiles) + (earlier waves) + (earlier steps) + prefix
    bool done = true;
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
      int before = 0, total = 0;
#pragma unroll
      for (int w = 0; w < BQ_WAVES; ++w) {
        const int v = s_cnt[par][w][c];
        before += w < wave ? v : 0;
        total += v;
      }
      int slot0 = cnt[c] + before;
      if (mine[c] && slot0 < nsample) {
#pragma unroll
        for (int s = 0; s < BQ_STEPS; ++s) {
          const unsigned long long m_ = mk[c][s];
          if (m_) {
            if (slot0 == 0) {   // the centre's very first hit is in this step (uniform)
              if (lane == 0) s_first[c] = tile0 + q0 + s * 64 + (__ffsll((long long)m_) - 1);
            }
            const int slot = slot0 + __popcll(m_ & lt_mask);
            if (((m_ >> lane) & 1ull) && slot < nsample)
              idx[(size_t)(j0 + c) * nsample + slot] = tile0 + q0 + s * 64 + lane;
            slot0 += __popcll(m_);
          }
        }
      }
      cnt[c] += total;
      done = done && (cnt[c] >= nsample);
    }
    if (done) break;   // uniform: cnt is the same in every thread
  }
  __syncthreads();
  // pad: first hit fills the unused slots; no hit => zeros (:34-38)
#pragma unroll
  for (int c = 0; c < CPW; ++c) {
    const int j = j0 + c;
    if (j < m && cnt[c] < nsample) {
      const int fill = cnt[c] == 0 ? 0 : s_first[c];
      for (int s = cnt[c] + t; s < nsample; s += BQ_THREADS) idx[(size_t)j * nsample + s] = fill;
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
  // centres per workgroup: keep >= ~2 workgroups per CU in flight
  const int total = m * b;
  if (total >= 2048) {
    hipLaunchKernelGGL(ball_query_kernel<4>, dim3(ceil_div(m, 4), b), dim3(BQ_THREADS), 0, s,
                       n, m, radius2, nsample, new_xyz, xyz, idx);
  } else if (total >= 1024) {
    hipLaunchKernelGGL(ball_query_kernel<2>, dim3(ceil_div(m, 2), b), dim3(BQ_THREADS), 0, s,
                       n, m, radius2, nsample, new_xyz, xyz, idx);
  } else {
    hipLaunchKernelGGL(ball_query_kernel<1>, dim3(m, b), dim3(BQ_THREADS), 0, s,
                       n, m, radius2, nsample, new_xyz, xyz, idx);
  }
  RFD_CHECK_LAUNCH();
  return 0;
}
