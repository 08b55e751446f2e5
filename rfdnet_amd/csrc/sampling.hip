// sampling.hip -- furthest point sampling + gather_points for gfx950.
//
// Replaces external/pointnet2_ops_lib/pointnet2_ops/_ext-src/src/sampling_gpu.cu
// (K1 furthest_point_sampling_kernel :69-173, K2 gather_points_kernel :8-20,
// K3 gather_points_grad_kernel :34-47).  Not a translation: the reference runs
// ONE 512-thread block per scene that re-streams all n points from L2/HBM in
// each of the m-1 serial rounds.  Here a scene is spread over G workgroups
// whose points (xyz, running min-distance, tie rank) live in REGISTERS for the
// whole kernel; a round is
//   local update + per-lane argmax  ->  wave argmax (DPP row/bcast reductions)
//   ->  workgroup argmax (LDS, one barrier)
//   ->  (G > 1) all-to-all exchange of the G candidates through 8-byte
//       {round-tag, value} granules in global memory (relaxed agent-scope
//       atomics; the data is its own flag) -> every wave picks the winner.
// A launch whose G workgroups are not all resident can never complete an exchange.  It does not hang: a wave
// that has polled for longer than the launch's time-out raises the region's sticky abort word, every workgroup
// (polling, or dispatched only later) sees it and leaves the round loop, and status bit 0 tells the host
// (the reference's launch that cannot run fails fast too, cuda_utils.h:30-39).
// HBM traffic is the algorithmic minimum: 12 B/point in, 4 B/point (temp) +
// 4 B/sample out.
//
// Bit-exactness with the CUDA kernel, including ties: the CUDA result is the
// maximum of d2 where equal values are ordered by (bit-reversed thread id
// k mod BS, then k / BS) -- thread-local strict '>' keeps the lowest k, the
// shared-memory tree keeps the left operand on ties (sampling_gpu.cu:59-65).
// We reduce on the 64-bit key (d2 bits << 32 | ~rank(k)) with
// rank(k) = bitrev(k mod BS) * ceil(n/BS) + k / BS, BS = opt_n_threads(n)
// (cuda_utils.h:13-19), which is a total order, so any reduction tree gives
// the CUDA answer.  d2 >= 0, so its bit pattern is monotone as unsigned.
#include "common.h"

namespace {

constexpr int FPS_THREADS = 256;
constexpr int FPS_WAVES = FPS_THREADS / 64;
constexpr unsigned FPS_SLOW_POLLS = 256;   // polls between two looks at the clock and the abort word (a healthy round
                                           // needs a handful of polls, so its path never pays for either)

typedef unsigned long long u64;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Wave-wide unsigned max with DPP (no LDS crossbar traffic): prefix max inside each 16-lane row (row_shr 1,2,4,8),
// row 0->1 / 2->3 (row_bcast:15), then 1->2,3 (row_bcast:31); lane 63 holds the result, read back through an SGPR.
// The DPP control sits ON the v_max_u32 (a lane whose source lane does not exist, or that the row mask excludes, keeps
// its value: bound_ctrl off), one instruction per step behind the two wait states a DPP read of a fresh VALU result
// needs.  Written out by hand: from `update_dpp` + max hipcc makes v_mov, s_nop, v_mov_dpp, v_max per step, and this
// chain runs two to four times in every one of the m serial rounds (round 5: see profiles/r05_fps_sweep.txt).
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
  asm volatile(
      "s_nop 1\n\t"
      "v_max_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(v));
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

struct Cand {
  u64 key;  // (d2 bits << 32) | ~rank, 0 = no valid point
  int k;
  float x, y, z;
};

// Broadcast the candidate of the (unique) lane holding the wave's maximum key.
// Two 32-bit DPP reductions (distance bits, then tie rank among the lanes that
// hold the maximum distance) + scalar read-lanes; everything ends up in SGPRs.
__device__ __forceinline__ Cand wave_select(const Cand &c) {
  const unsigned hi = (unsigned)(c.key >> 32), lo = (unsigned)c.key;
  const unsigned mh = wave_max_u32(hi);
  const unsigned ml = wave_max_u32(hi == mh ? lo : 0u);
  const u64 m = __ballot(hi == mh && lo == ml);
  const int src = __builtin_amdgcn_readfirstlane(__ffsll((long long)m) - 1);
  Cand r;
  r.key = ((u64)mh << 32) | ml;
  r.k = __builtin_amdgcn_readlane(c.k, src);
  r.x = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, c.x), src));
  r.y = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, c.y), src));
  r.z = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, c.z), src));
  return r;
}

__device__ __forceinline__ unsigned fps_rank(int k, int bs_log2, int cpb) {
  const unsigned tid = (unsigned)k & ((1u << bs_log2) - 1u);
  const unsigned rev = bs_log2 ? (__brev(tid) >> (32 - bs_log2)) : 0u;
  return rev * (unsigned)cpb + ((unsigned)k >> bs_log2);
}

// MULTI = false: one workgroup holds the scene (no exchange code at all in the instantiation).
template <int PPT, bool MULTI>
__global__ __launch_bounds__(FPS_THREADS) void fps_kernel(
    int n, int m, int G, int Gw, int bs_log2, int cpb, const float *__restrict__ dataset,
    float *__restrict__ temp, int *__restrict__ idxs,
    float *__restrict__ new_xyz, u64 *region, unsigned *status, u64 timeout_ticks) {
  const int batch = blockIdx.x / G;
  const int g = blockIdx.x - batch * G;
  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  dataset += (size_t)batch * n * 3;
  temp += (size_t)batch * n;
  idxs += (size_t)batch * m;
  if (new_xyz) new_xyz += (size_t)batch * m * 3;
  u64 *abort_word = region;  // one per launch: any scene's time-out ends them all
  // Gw = exchange units a round waits for: G, or more when a test asks for units that never publish
  // (rfd_fps_test_phantom_units: the deterministic way into the time-out path)
  u64 *slots = region + FPS_REGION_HEAD + (size_t)batch * Gw * 10;  // [g][parity][5]

  __shared__ int s_abort;
  // a wave's candidate: {key lo, key hi, k, x} and {y, z}; [parity][wave]
  __shared__ __attribute__((aligned(16))) u32x4 s_a[2][FPS_WAVES];
  __shared__ __attribute__((aligned(8))) f32x2 s_yz[2][FPS_WAVES];
  // the round's global winner, written by the polling wave: same layout; [parity]
  __shared__ __attribute__((aligned(16))) u32x4 s_win[2];
  __shared__ __attribute__((aligned(8))) f32x2 s_winyz[2];

  // ---- load this thread's points once; they stay in registers ----
  float px[PPT], py[PPT], pz[PPT], td[PPT];
  unsigned nrank[PPT];  // ~rank, 0 = point absent or skipped
  const int base = g * (PPT * FPS_THREADS) + t;
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int k = base + i * FPS_THREADS;
    const bool in = k < n;
    px[i] = in ? dataset[(size_t)k * 3 + 0] : 0.f;
    py[i] = in ? dataset[(size_t)k * 3 + 1] : 0.f;
    pz[i] = in ? dataset[(size_t)k * 3 + 2] : 0.f;
    const float mag = sumsq3(px[i], py[i], pz[i]);       // sampling_gpu.cu:100
    const bool skip = !in || ((double)mag <= 1e-3);      // :101 (double literal)
    nrank[i] = skip ? 0u : ~fps_rank(k, bs_log2, cpb);
    // running min-distance, 1e10 at the start (sampling.cpp:74-76).  A skipped / absent point carries 0 instead:
    // min(d, 0) = 0 and rank 0 make its key 0 = "no candidate" with no per-point predicate in the round loop
    // (a real point at distance 0 still has a non-zero rank); the reference never touches a skipped point's
    // temp, so 1e10 is what is written back for it at the end
    td[i] = skip ? 0.f : 1e10f;
  }
  const float p0x = dataset[0], p0y = dataset[1], p0z = dataset[2];

  float cx = p0x, cy = p0y, cz = p0z;  // old = 0 (:86)
  if (g == 0 && t == 0) {
    idxs[0] = 0;
    if (new_xyz) { new_xyz[0] = p0x; new_xyz[1] = p0y; new_xyz[2] = p0z; }
  }
  if (t == 0) s_abort = 0;  // (the first barrier of round 1 orders it)

  for (int j = 1; j < m; ++j) {
    const int par = j & 1;
    // ---- local update + argmax over this thread's points ----
    Cand c;
    c.key = 0; c.k = 0; c.x = p0x; c.y = p0y; c.z = p0z;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const float d = sumsq3(px[i] - cx, py[i] - cy, pz[i] - cz);  // :103-104
      const float d2 = fminf(d, td[i]);                             // :106
      td[i] = d2;
      const u64 key = ((u64)__float_as_uint(d2) << 32) | nrank[i];
      const bool better = key > c.key;
      c.key = better ? key : c.key;
      c.k = better ? base + i * FPS_THREADS : c.k;
      c.x = better ? px[i] : c.x;
      c.y = better ? py[i] : c.y;
      c.z = better ? pz[i] : c.z;
    }
    // ---- wave argmax, then workgroup argmax through LDS ----
    Cand w = wave_select(c);
    if (lane == 0) {
      s_a[par][wave] = u32x4{(unsigned)w.key, (unsigned)(w.key >> 32), (unsigned)w.k, __float_as_uint(w.x)};
      s_yz[par][wave] = f32x2{w.y, w.z};
    }
    __syncthreads();
    Cand b;
    b.key = 0; b.k = 0; b.x = p0x; b.y = p0y; b.z = p0z;
    if (!MULTI || wave == 0) {
      // all four candidates in ONE round of broadcast reads (left to itself hipcc makes every payload read
      // conditional on the key comparison before it: four dependent LDS round trips per round), then register selects
      u32x4 ca[FPS_WAVES];
      f32x2 cyz[FPS_WAVES];
#pragma unroll
      for (int q = 0; q < FPS_WAVES; ++q) { ca[q] = s_a[par][q]; cyz[q] = s_yz[par][q]; }
#pragma unroll
      for (int q = 0; q < FPS_WAVES; ++q) asm volatile("" : "+v"(ca[q]), "+v"(cyz[q]));
      b.key = ((u64)ca[0].y << 32) | ca[0].x; b.k = (int)ca[0].z;
      b.x = __uint_as_float(ca[0].w); b.y = cyz[0].x; b.z = cyz[0].y;
#pragma unroll
      for (int q = 1; q < FPS_WAVES; ++q) {
        const u64 kq = ((u64)ca[q].y << 32) | ca[q].x;
        const bool better = kq > b.key;
        b.key = better ? kq : b.key;
        b.k = better ? (int)ca[q].z : b.k;
        b.x = better ? __uint_as_float(ca[q].w) : b.x;
        b.y = better ? cyz[q].x : b.y;
        b.z = better ? cyz[q].y : b.z;
      }
    }
    if constexpr (MULTI) {
      // ONE wave of the workgroup runs the exchange; the other three sleep at the barrier below.  (Round 5, GPU call
      // 11: four pollers per CU queue behind each other in the CU's memory path -- the guide's hand-off table: 0.8 us
      // idle, 1.2-1.5 us with 2-5 waves streaming on the endpoint CU -- 5.19 -> 4.95 ms at SA1 for one poller + a second
      // barrier.)
      if (wave == 0) {
        // ---- publish this workgroup's candidate: 5 tagged granules ----
        u64 *mine = slots + ((size_t)g * 2 + par) * 5;
        if (t < 5) {
          unsigned payload = (unsigned)(b.key >> 32);
          payload = t == 1 ? (b.key ? (unsigned)b.k : 0xffffffffu) : payload;
          payload = t == 2 ? __float_as_uint(b.x) : payload;
          payload = t == 3 ? __float_as_uint(b.y) : payload;
          payload = t == 4 ? __float_as_uint(b.z) : payload;
          __hip_atomic_store(mine + t, ((u64)(unsigned)j << 32) | payload,
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // ---- gather all G candidates (lane = workgroup) ----
        const u64 *theirs = slots + ((size_t)lane * 2 + par) * 5;
        unsigned f0 = 0, f1 = 0, f2 = 0, f3 = 0, f4 = 0;
        unsigned spins = 0;
        u64 t_first = 0;
        for (;;) {
          bool ok = true;
          if (lane < Gw) {
            const u64 g0 = __hip_atomic_load(theirs + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const u64 g1 = __hip_atomic_load(theirs + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const u64 g2 = __hip_atomic_load(theirs + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const u64 g3 = __hip_atomic_load(theirs + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const u64 g4 = __hip_atomic_load(theirs + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned tag = (unsigned)j;
            ok = (unsigned)(g0 >> 32) == tag && (unsigned)(g1 >> 32) == tag &&
                 (unsigned)(g2 >> 32) == tag && (unsigned)(g3 >> 32) == tag &&
                 (unsigned)(g4 >> 32) == tag;
            f0 = (unsigned)g0; f1 = (unsigned)g1; f2 = (unsigned)g2;
            f3 = (unsigned)g3; f4 = (unsigned)g4;
          }
          if (__all(ok)) break;
          if ((++spins % FPS_SLOW_POLLS) == 0) {
            // slow path: somebody is late.  Leave -- for good, the whole launch -- when another workgroup has
            // already given up or this wave has waited out the time-out itself (wall clock, not a poll count: a poll
            // takes anything from 0.3 to several us depending on what else uses the memory system).
            const u64 now = (u64)wall_clock64();
            if (t_first == 0) t_first = now;
            const bool dead = __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
            if (dead || now - t_first > timeout_ticks) {
              if (lane == 0) {
                __hip_atomic_store(abort_word, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                atomicOr(status, 1u);
                s_abort = 1;
              }
              break;  // this round's "winner" is garbage; everybody reads the flag behind the barrier below
            }
          }
          __builtin_amdgcn_s_sleep(1);
        }
        // a workgroup without a valid point publishes k = -1; rank is recomputed
        // from k so ties between workgroups order exactly as in the CUDA tree
        Cand o;
        const bool valid = lane < G && (int)f1 >= 0;
        o.key = valid ? (((u64)f0 << 32) | (u64)(~fps_rank((int)f1, bs_log2, cpb))) : 0ull;
        o.k = (int)f1;
        o.x = __uint_as_float(f2);
        o.y = __uint_as_float(f3);
        o.z = __uint_as_float(f4);
        b = wave_select(o);
        if (lane == 0) {
          s_win[par] = u32x4{(unsigned)b.key, (unsigned)(b.key >> 32), (unsigned)b.k, __float_as_uint(b.x)};
          s_winyz[par] = f32x2{b.y, b.z};
        }
      }
      __syncthreads();
      u32x4 wa = s_win[par];
      f32x2 wyz = s_winyz[par];
      int gave_up = s_abort;
      asm volatile("" : "+v"(wa), "+v"(wyz), "+v"(gave_up));
      if (gave_up) break;      // the polling wave gave up (time-out / the launch's abort word): everybody leaves, together
      b.key = ((u64)wa.y << 32) | wa.x; b.k = (int)wa.z;
      b.x = __uint_as_float(wa.w); b.y = wyz.x; b.z = wyz.y;
    }
    if (b.key == 0ull) { b.k = 0; b.x = p0x; b.y = p0y; b.z = p0z; }  // all skipped
    cx = b.x; cy = b.y; cz = b.z;  // old = dists_i[0] (:170)
    if (g == 0 && t == 0) {
      idxs[j] = b.k;
      if (new_xyz) { new_xyz[j * 3 + 0] = b.x; new_xyz[j * 3 + 1] = b.y; new_xyz[j * 3 + 2] = b.z; }
    }
  }
  // the CUDA kernel leaves the final min-distances in temp
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int k = base + i * FPS_THREADS;
    if (k < n) temp[k] = nrank[i] ? td[i] : 1e10f;
  }
}

__global__ void gather_points_kernel(int c, int n, int m,
                                     const float *__restrict__ points,
                                     const int *__restrict__ idx,
                                     float *__restrict__ out) {
  // grid (ceil(m/256), c, b): out[b,l,j] = points[b,l,idx[b,j]]
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int l = blockIdx.y, bi = blockIdx.z;
  if (j >= m) return;
  const int a = idx[(size_t)bi * m + j];
  out[((size_t)bi * c + l) * m + j] = points[((size_t)bi * c + l) * n + a];
}

__global__ void gather_points_grad_kernel(int c, int n, int m,
                                          const float *__restrict__ grad_out,
                                          const int *__restrict__ idx,
                                          float *__restrict__ grad_points) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int l = blockIdx.y, bi = blockIdx.z;
  if (j >= m) return;
  const int a = idx[(size_t)bi * m + j];
  atomicAdd(grad_points + ((size_t)bi * c + l) * n + a,
            grad_out[((size_t)bi * c + l) * m + j]);
}

#ifndef RFD_NO_TEST_HOOKS
// Test hook (rfd_test_hold_cus): one 64-thread workgroup per CU, each holding the CU's whole LDS, until *release != 0
// or max_ticks of wall clock have passed -- so that nothing that needs LDS can be placed beside it.
constexpr int CU_LDS_BYTES = 160 * 1024;      // gfx950: the whole LDS of a compute unit
__global__ void hold_cus_kernel(const unsigned *release, u64 max_ticks) {
  __shared__ unsigned char held[CU_LDS_BYTES];
  if (threadIdx.x == 0) {
    reinterpret_cast<volatile unsigned char *>(held)[CU_LDS_BYTES - 1] = 1;      // keep the allocation
    const u64 t0 = (u64)wall_clock64();
    while (__hip_atomic_load(release, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u &&
           (u64)wall_clock64() - t0 < max_ticks)
      __builtin_amdgcn_s_sleep(32);
  }
}
#endif  // RFD_NO_TEST_HOOKS

template <int PPT, bool MULTI>
int launch_fps(int nb, int n, int m, int G, int Gw, int bs_log2, int cpb,
               const float *dataset, float *temp, int *idxs, float *new_xyz,
               u64 *region, unsigned *status, u64 timeout_ticks, hipStream_t s) {
  // (packing the G exchanging workgroups onto one XCD -- launching 8x the blocks
  // and using every 8th -- was measured 15 % SLOWER than letting them spread)
  hipLaunchKernelGGL((fps_kernel<PPT, MULTI>), dim3(nb * G), dim3(FPS_THREADS), 0, s, n,
                     m, G, Gw, bs_log2, cpb, dataset, temp, idxs, new_xyz, region,
                     status, timeout_ticks);
  RFD_CHECK_LAUNCH();
  return 0;
}

int fps_impl(int b, int n, int m, const float *dataset, float *temp, int *idxs,
             float *new_xyz, void *stream) {
  if (m <= 0 || b <= 0) return 0;  // sampling_gpu.cu:73
  if (n <= 0) { rfd_set_error("furthest_point_sampling: n <= 0", hipErrorInvalidValue); return (int)hipErrorInvalidValue; }
  hipStream_t s = (hipStream_t)stream;
  RfdWorkspace *ws;
  int rc = rfd_get_workspace(&ws);
  if (rc) return rc;
  // geometry: G workgroups x 256 threads x PPT points cover n
  const int per_thread = ceil_div(n, FPS_THREADS);
  const int forced = ws->fps_force_ppt.load(std::memory_order_relaxed);
  int ppt, G;
  bool chosen = false;
  if (forced && per_thread > forced) {  // rfd_fps_set_geometry: sweeps and tests
    ppt = forced;
    G = ceil_div(n, FPS_THREADS * ppt);
    // a forced size that still leaves ONE workgroup (4096 < n <= 256 ppt) has no single-workgroup instantiation
    // above 16 points per thread: the hook then does not apply ("results never depend on it"), the automatic
    // geometry below does
    chosen = G > 1;
  }
  if (chosen) {
  } else if (per_thread <= 16) {  // one workgroup holds the scene
    G = 1;
    ppt = per_thread <= 1 ? 1 : per_thread <= 2 ? 2 : per_thread <= 4 ? 4 : per_thread <= 8 ? 8 : 16;
  } else {
    // ~10 points/thread (SA1: 32 exchange units; the sweep towards fewer, fatter units is profiles/r05_fps_sweep.txt);
    // more when the scene would need > 64 workgroups
    ppt = 10;
    G = ceil_div(n, FPS_THREADS * ppt);
    if (G > 64) { ppt = 16; G = ceil_div(n, FPS_THREADS * ppt); }
    if (G > 64) { ppt = 32; G = ceil_div(n, FPS_THREADS * ppt); }
    if (G > 64) { ppt = 64; G = ceil_div(n, FPS_THREADS * ppt); }
  }
  if (G > 64) { rfd_set_error("furthest_point_sampling: n > 1048576 unsupported", hipErrorInvalidValue); return (int)hipErrorInvalidValue; }
  const int bs = ref_opt_n_threads(n);
  int bs_log2 = 0;
  while ((1 << bs_log2) < bs) ++bs_log2;
  const int cpb = ceil_div(n, bs);
  int Gw = G;
  if (G > 1) {
#ifndef RFD_NO_TEST_HOOKS
    // one-shot: the hook arms the NEXT multi-workgroup call only, a forgotten reset cannot poison a process
    const int phantom = ws->fps_test_phantom.exchange(0);
    Gw = G + phantom > 64 ? 64 : G + phantom;
#endif
  }
  const int batches_per_launch = G > 1 ? (FPS_MAX_WG / Gw) : b;
  const u64 timeout_ticks = (u64)ws->fps_timeout_ms.load(std::memory_order_relaxed) * (u64)ws->wall_clock_khz;
  unsigned *status = rfd_status_word(ws, s);
  for (int b0 = 0; b0 < b; b0 += batches_per_launch) {
    const int nb = (b - b0) < batches_per_launch ? (b - b0) : batches_per_launch;
    u64 *region = nullptr;
    if (G > 1) {
      // this stream's own region (launches on a stream are serial); abort word + exchange granules zeroed per launch
      region = rfd_fps_region(ws, s);
      RFD_CHECK(hipMemsetAsync(region, 0, sizeof(u64) * (FPS_REGION_HEAD + (size_t)nb * Gw * 10), s));
    }
    const float *ds = dataset + (size_t)b0 * n * 3;
    float *tp = temp + (size_t)b0 * n;
    int *ix = idxs + (size_t)b0 * m;
    float *nx = new_xyz ? new_xyz + (size_t)b0 * m * 3 : nullptr;
#define FPS_CASE(P, M) case P: rc = launch_fps<P, M>(nb, n, m, G, Gw, bs_log2, cpb, ds, tp, ix, nx, region, status, timeout_ticks, s); break;
    if (G > 1) {
      switch (ppt) {
        FPS_CASE(5, true) FPS_CASE(8, true) FPS_CASE(10, true) FPS_CASE(16, true) FPS_CASE(20, true) FPS_CASE(32, true)
        FPS_CASE(40, true) FPS_CASE(64, true)
        default: rc = (int)hipErrorInvalidValue;
      }
    } else {
      switch (ppt) {
        FPS_CASE(1, false) FPS_CASE(2, false) FPS_CASE(4, false) FPS_CASE(8, false) FPS_CASE(16, false)
        default: rc = (int)hipErrorInvalidValue;
      }
    }
#undef FPS_CASE
    if (rc) return rc;
  }
  return 0;
}

}  // namespace

RFD_API int furthest_point_sampling_kernel_wrapper(int b, int n, int m,
                                                   const float *dataset,
                                                   float *temp, int *idxs,
                                                   void *stream) {
  return fps_impl(b, n, m, dataset, temp, idxs, nullptr, stream);
}

RFD_API int rfd_furthest_point_sampling_gather(int b, int n, int m,
                                               const float *dataset,
                                               float *temp, int *idxs,
                                               float *new_xyz, void *stream) {
  return fps_impl(b, n, m, dataset, temp, idxs, new_xyz, stream);
}

#ifndef RFD_NO_TEST_HOOKS
// Test hook: occupy all but `leave_free_cus` compute units of the current device with workgroups that hold a CU's
// whole LDS each, until *release_flag (device memory) becomes non-zero or max_ms have passed (hard upper bound: the
// hook can never hang a device).  A multi-workgroup FPS launched beside it then CANNOT have all its workgroups
// resident -- the situation rfd_fps_set_timeout_ms exists for (tests/test_gpu_fps_abort.py).  Returns the number of
// holding workgroups (> 0) or a negative hipError.
RFD_API int rfd_test_hold_cus(int leave_free_cus, const unsigned *release_flag, int max_ms, void *stream) {
  RfdWorkspace *ws;
  int rc = rfd_get_workspace(&ws);
  if (rc) return -rc;
  const int n = ws->num_cu - leave_free_cus;
  if (!release_flag || leave_free_cus < 0 || n <= 0 || max_ms <= 0 || max_ms > 10000) {
    rfd_set_error("rfd_test_hold_cus: arguments", hipErrorInvalidValue);
    return -(int)hipErrorInvalidValue;
  }
  // (static LDS of exactly one CU's worth: the device attribute MaxSharedMemoryPerMultiprocessor reports 64 KiB on this
  // stack, and two holders then shared a CU and left half the chip free)
  hipLaunchKernelGGL(hold_cus_kernel, dim3(n), dim3(64), 0, (hipStream_t)stream, release_flag,
                     (u64)max_ms * (u64)ws->wall_clock_khz);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { rfd_set_error("rfd_test_hold_cus", e); return -(int)e; }
  return n;
}
#endif  // RFD_NO_TEST_HOOKS

RFD_API int gather_points_kernel_wrapper(int b, int c, int n, int npoints,
                                         const float *points, const int *idx,
                                         float *out, void *stream) {
  if (b <= 0 || c <= 0 || npoints <= 0) return 0;
  hipLaunchKernelGGL(gather_points_kernel, dim3(ceil_div(npoints, 256), c, b),
                     dim3(256), 0, (hipStream_t)stream, c, n, npoints, points,
                     idx, out);
  RFD_CHECK_LAUNCH();
  return 0;
}

RFD_API int gather_points_grad_kernel_wrapper(int b, int c, int n, int npoints,
                                              const float *grad_out,
                                              const int *idx,
                                              float *grad_points,
                                              void *stream) {
  if (b <= 0 || c <= 0 || npoints <= 0) return 0;
  hipLaunchKernelGGL(gather_points_grad_kernel,
                     dim3(ceil_div(npoints, 256), c, b), dim3(256), 0,
                     (hipStream_t)stream, c, n, npoints, grad_out, idx,
                     grad_points);
  RFD_CHECK_LAUNCH();
  return 0;
}
