// mise.hip -- batched multiresolution iso-surface extraction state machine and
// dense query-grid generation for gfx950.
//
// Replaces external/libmise/mise.pyx:33-369 (Cython octree, one proposal at a
// time on the CPU, std::map point hash) and the query-point arithmetic of
// Generator3D.generate_from_latent (models/iscnet/modules/generator.py:91-115)
// and make_3d_grid (external/common.py:157-176).
//
// MI355X-first restatement: the state of all K proposals is dense in HBM
// (a byte per lattice point, a byte per octree voxel per level, 4 B per value),
// one launch advances every proposal by one round, and nothing crosses PCIe
// except one 4-byte "points left" counter per round.  These kernels are pure
// byte/flag traffic: HBM-bound, coalesced along z (the fastest lattice axis).
//
// Equivalence with the octree (argued in DESIGN.md "MISE"): a leaf voxel V of
// size s is marked by exactly the known grid points of its closed cube
// [lo, lo+s]^3 (mise.pyx:205-228 marks, for every known point, the leaf that
// contains each of the 8 unit cells around it), marking is complete before any
// subdivision (:230-251) and children created in a pass are not examined in
// that pass -- reproduced by processing levels from fine to coarse.
#include "common.h"
#include "../../include/rfd_occ.h"

namespace {

__host__ __device__ inline size_t cube(size_t n) { return n * n * n; }

// offset (in elements) of level l inside a proposal's vstate block
__host__ __device__ inline size_t vstate_offset(int res0, int l) {
  size_t o = 0;
  for (int i = 0; i < l; ++i) o += cube((size_t)res0 << i);
  return o;
}

__global__ void grid_points_kernel(int n, float lo, float hi, float scale,
                                   float *__restrict__ pts, int n_padded) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_padded) return;
  const int total = n * n * n;
  float x = 0.f, y = 0.f, z = 0.f;
  if (e < total) {
    const int k = e % n, j = (e / n) % n, i = e / (n * n);
    const float step = n > 1 ? (hi - lo) / (float)(n - 1) : 0.f;
    const int half = n / 2;
    // torch.linspace (GPU formula): symmetric evaluation from both ends
    auto ax = [&](int q) { return q < half ? lo + step * (float)q : hi - step * (float)(n - 1 - q); };
    x = scale * ax(i);  // box_size * grid (generator.py:92-94)
    y = scale * ax(j);
    z = scale * ax(k);
  }
  pts[(size_t)e * 3 + 0] = x;
  pts[(size_t)e * 3 + 1] = y;
  pts[(size_t)e * 3 + 2] = z;
}

// pstate: 1 on the level-0 lattice (multiples of 2^depth), else 0 (mise.pyx:72-85): the array
// is cleared by a memset, this kernel sets the (res0 + 1)^3 lattice points of every proposal
// (a thread per byte of the whole array, with three 64-bit divisions each, took 177 us).
__global__ void mise_init_points_kernel(int R1, int vs0, int L, size_t n_per,
                                        unsigned char *__restrict__ pstate) {
  const unsigned e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (unsigned)(L * L * L)) return;
  const unsigned k = e % L, j = (e / L) % L, i = e / (L * L);
  pstate[(size_t)blockIdx.y * n_per + ((size_t)(i * vs0) * R1 + j * vs0) * R1 + k * vs0] = 1;
}

__global__ void mise_init_voxels_kernel(size_t n0, size_t n_per,
                                        unsigned char *__restrict__ vstate) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_per) return;
  vstate[(size_t)blockIdx.y * n_per + e] = e < n0 ? 1 : 0;  // level 0: all leaves
}

// ---- 16 bytes of point state per thread ------------------------------------------
// The state arrays are scanned as one flat byte stream of K*n_per bytes in
// aligned 16-byte chunks (n_per = (R+1)^3 is odd, so a chunk may straddle two
// proposals: `split` = number of its bytes that belong to the first one).
struct Chunk {
  unsigned w[4];   // the 16 state bytes (0 beyond the end of the array)
  int k0;          // proposal of byte 0
  int split;       // bytes [0,split) -> k0, [split,16) -> k0 + 1
  size_t e0;       // lattice index (within k0) of byte 0
};

__device__ __forceinline__ Chunk load_chunk(const unsigned char *__restrict__ ps, size_t total,
                                            size_t n_per, size_t chunk) {
  Chunk c;
  const size_t b0 = chunk * 16;
  if (b0 + 16 <= total) {
    const uint4 v = *reinterpret_cast<const uint4 *>(ps + b0);
    c.w[0] = v.x; c.w[1] = v.y; c.w[2] = v.z; c.w[3] = v.w;
  } else {
    c.w[0] = c.w[1] = c.w[2] = c.w[3] = 0;
    for (int j = 0; j < 16; ++j)
      if (b0 + j < total) c.w[j >> 2] |= (unsigned)ps[b0 + j] << (8 * (j & 3));
  }
  c.k0 = (int)(b0 / n_per);
  c.e0 = b0 - (size_t)c.k0 * n_per;
  const size_t left = n_per - c.e0;
  c.split = left < 16 ? (int)left : 16;
  return c;
}

// bit j set <=> state byte j == 1 (exists, unknown).  States are 0..3, so
// (x | x>>1) & 1 tests "byte != 0" without cross-byte carries.
__device__ __forceinline__ unsigned unknown_mask16(const Chunk &c) {
  unsigned m = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const unsigned x = c.w[q] ^ 0x01010101u;
    const unsigned z = ~(x | (x >> 1)) & 0x01010101u;   // 1 where byte == 1
    // gather the four flag bits (bits 0, 8, 16, 24) into a nibble
    const unsigned nib = (z & 1u) | ((z >> 7) & 2u) | ((z >> 14) & 4u) | ((z >> 21) & 8u);
    m |= nib << (4 * q);
  }
  return m;
}

__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// Unknown points (state == 1) per proposal.  grid (COUNT_PARTS, K): a workgroup strides over
// the aligned 16-byte words of ONE proposal with four loads in flight per thread; the few bytes
// before / after the aligned span are counted by the first workgroup.  (One 16-byte chunk per
// thread over the flat array, with a 64-bit division each, ran at 0.85 TB/s.)
constexpr int COUNT_PARTS = 8;

__device__ __forceinline__ int ones16(const uint4 v) {
  int c = 0;
  const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const unsigned x = w[q] ^ 0x01010101u;
    c += __popc(~(x | (x >> 1)) & 0x01010101u);   // states are 0..3: byte == 1
  }
  return c;
}

__global__ __launch_bounds__(256) void mise_count_kernel(size_t n_per,
                                                         const unsigned char *__restrict__ pstate,
                                                         int *__restrict__ counts) {
  __shared__ int wsum[4];
  const int k = blockIdx.y;
  const unsigned char *base = pstate + (size_t)k * n_per;
  const unsigned head = (unsigned)((16 - (reinterpret_cast<uintptr_t>(base) & 15)) & 15);
  const unsigned nvec = (unsigned)((n_per - head) / 16);
  const uint4 *vec = reinterpret_cast<const uint4 *>(base + head);
  int cnt = 0;
  const unsigned stride = COUNT_PARTS * 256;
  unsigned v = blockIdx.x * 256 + threadIdx.x;
  for (; v + 3 * stride < nvec; v += 4 * stride) {
    const uint4 a = vec[v], b = vec[v + stride], c = vec[v + 2 * stride], d = vec[v + 3 * stride];
    cnt += ones16(a) + ones16(b) + ones16(c) + ones16(d);
  }
  for (; v < nvec; v += stride) cnt += ones16(vec[v]);
  if (blockIdx.x == 0) {
    const unsigned tail0 = head + nvec * 16;
    if (threadIdx.x < head) cnt += base[threadIdx.x] == 1;
    if (tail0 + threadIdx.x < n_per && threadIdx.x < 16) cnt += base[tail0 + threadIdx.x] == 1;
  }
  cnt = wave_sum(cnt);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int tot = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    if (tot) atomicAdd(counts + k, tot);
  }
}

__global__ __launch_bounds__(256) void mise_collect_kernel(
    int R1, size_t total, size_t n_per, int K, const unsigned char *__restrict__ pstate,
    const int *__restrict__ offsets, int *__restrict__ cursors, float box_size,
    float *__restrict__ pts, int *__restrict__ lin) {
  const size_t chunk = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  unsigned m = 0, lo = 0xffffu;
  int k0 = -1;
  size_t e0 = 0;
  if (chunk * 16 < total) {
    const Chunk c = load_chunk(pstate, total, n_per, chunk);
    m = unknown_mask16(c);
    lo = c.split >= 16 ? 0xffffu : ((1u << c.split) - 1u);
    k0 = c.k0;
    e0 = c.e0;
  }
  const int c0 = __popc(m & lo), c1 = __popc(m & ~lo);
  if (!__any(c0 | c1)) return;
  int base0 = 0, base1 = 0;
  const int kf = __shfl(k0, 0);
  if (__all((k0 == kf || (c0 | c1) == 0) && c1 == 0)) {
    // one atomic per wave + an in-wave exclusive prefix
    int incl = c0;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int o = __shfl_up(incl, off);
      if (lane >= off) incl += o;
    }
    const int tot = __shfl(incl, 63);
    int wb = 0;
    if (lane == 0) wb = atomicAdd(cursors + kf, tot);
    wb = __shfl(wb, 0);
    base0 = wb + incl - c0;
  } else {                                   // a proposal boundary inside the wave
    if (c0) base0 = atomicAdd(cursors + k0, c0);
    if (c1 && k0 + 1 < K) base1 = atomicAdd(cursors + k0 + 1, c1);
  }
  const float res = (float)(R1 - 1);
  unsigned mm = m;
  while (mm) {
    const int j = __ffs(mm) - 1;
    mm &= mm - 1;
    const bool second = !((lo >> j) & 1u);
    const int k = second ? k0 + 1 : k0;
    if (k >= K) break;
    const unsigned e = (unsigned)(second ? (e0 + j - n_per) : (e0 + j));      // < n_per < 2^31
    const int slot = offsets[k] + (second ? base1++ : base0++);
    const unsigned row = e / (unsigned)R1;                                      // 32-bit divisions
    const int kz = (int)(e - row * (unsigned)R1), ix = (int)(row / (unsigned)R1),
              jy = (int)(row - (unsigned)ix * (unsigned)R1);
    // pointsf = points / resolution; box_size * (pointsf - 0.5)  (generator.py:106-109)
    pts[(size_t)slot * 3 + 0] = box_size * ((float)ix / res - 0.5f);
    pts[(size_t)slot * 3 + 1] = box_size * ((float)jy / res - 0.5f);
    pts[(size_t)slot * 3 + 2] = box_size * ((float)kz / res - 0.5f);
    lin[slot] = (int)e;
  }
}

__global__ void mise_scatter_kernel(size_t n_per, const int *__restrict__ tile_prop,
                                    const int *__restrict__ tile_src,
                                    const int *__restrict__ lin,
                                    const float *__restrict__ logits,
                                    float *__restrict__ values,
                                    unsigned char *__restrict__ pstate) {
  const int tile = blockIdx.x;
  const int k = tile_prop[tile];
  if (k < 0) return;
  const size_t slot = (size_t)tile * RFD_OCC_TILE + threadIdx.x;
  // lin may be shared between proposals (tile_src: the tile of `lin` this tile's points come from)
  const int l = lin[tile_src ? (size_t)tile_src[tile] * RFD_OCC_TILE + threadIdx.x : slot];
  if (l < 0) return;
  values[(size_t)k * n_per + l] = logits[slot];  // mise.pyx:101
  pstate[(size_t)k * n_per + l] = 2;             // :102 known = True
}

// ---- dirty slabs (round 6) ----------------------------------------------------------------------------------
// Whether a leaf voxel splits is a function of the KNOWN points of its closed cube.  After a pass no leaf it examined is
// mixed (it would have been split), so in the next pass only two kinds of leaf can split: one whose cube holds a point
// that has become known SINCE -- a query point of the round just decoded -- and one the previous pass CREATED (children
// are not examined in the pass that creates them, mise.pyx:239-251).  The last rounds of an octree evaluate a few
// thousand points, yet a pass stages every slab that holds a leaf: 2.6 ms per tail round at 128^3, 5 % of a scene.
// A byte per (proposal, level, slab of the LDS kernel's decomposition): the query list of the round marks the slabs of
// the (up to eight) voxels per level whose cube contains each point, a subdivision marks its children's slabs in the
// buffer of the NEXT pass, and a pass told to (`use_dirty`: the caller does so for sparse rounds) skips clean slabs.
// Results are identical by construction; tests/test_gpu_generator.py runs every pass of every case in this mode
// against the octree oracle.
constexpr int SUB_TJ = 8, SUB_TK = 32;

__host__ __device__ inline int sub_tiles_k(int nl) { return (nl + SUB_TK - 1) / SUB_TK; }
__host__ __device__ inline int sub_tiles_j(int nl) { return (nl + SUB_TJ - 1) / SUB_TJ; }
__host__ __device__ inline size_t dirty_offset(int res0, int l) {
  size_t o = 0;
  for (int i = 0; i < l; ++i) {
    const int nl = res0 << i;
    o += (size_t)sub_tiles_k(nl) * sub_tiles_j(nl) * nl;
  }
  return o;
}
__device__ inline size_t dirty_tile(int nl, int vi, int vj, int vk) {
  return ((size_t)vi * sub_tiles_j(nl) + vj / SUB_TJ) * sub_tiles_k(nl) + vk / SUB_TK;
}
// the eight children of voxel (vi, vj, vk) of level l live in two slabs of level l + 1 (2 vj and 2 vk are even: one j / k group)
__device__ inline void mark_children(unsigned char *dirty_next, int res0, int l, int nl, int vi, int vj, int vk) {
  unsigned char *d = dirty_next + dirty_offset(res0, l + 1);
  d[dirty_tile(2 * nl, 2 * vi, 2 * vj, 2 * vk)] = 1;
  d[dirty_tile(2 * nl, 2 * vi + 1, 2 * vj, 2 * vk)] = 1;
}

// one thread per query slot of the round just decoded
__global__ void mise_mark_dirty_kernel(size_t n_slots, int R1, int res0, int depth, const int *__restrict__ lin,
                                       const int *__restrict__ tile_prop, size_t d_per,
                                       unsigned char *__restrict__ dirty) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_slots) return;
  const int p = lin[i];
  if (p < 0) return;
  const int kp = tile_prop[i / RFD_OCC_TILE];
  if (kp < 0) return;
  const int c[3] = {p / (R1 * R1), (p / R1) % R1, p % R1};
  unsigned char *dk = dirty + (size_t)kp * d_per;
  for (int l = 0; l < depth; ++l) {
    const int s = 1 << (depth - l), nl = res0 << l;
    int v[3][2], n[3];
    for (int a = 0; a < 3; ++a) {              // the voxels of this level whose closed interval holds c[a]
      const int q = c[a] / s;
      n[a] = 0;
      if (q < nl) v[a][n[a]++] = q;
      if (c[a] % s == 0 && q > 0) v[a][n[a]++] = q - 1;
    }
    unsigned char *d = dk + dirty_offset(res0, l);
    for (int a = 0; a < n[0]; ++a)
      for (int b = 0; b < n[1]; ++b)
        for (int e = 0; e < n[2]; ++e) d[dirty_tile(nl, v[0][a], v[1][b], v[2][e])] = 1;
  }
}

// One level of subdivide_voxels.  Thread per voxel of level `l`.
__global__ void mise_subdivide_kernel(int R1, int res0, int depth, int l, double thr,
                                      size_t n_per, size_t v_per,
                                      const float *__restrict__ values,
                                      unsigned char *__restrict__ pstate,
                                      unsigned char *__restrict__ vstate,
                                      const int *__restrict__ evaluated, size_t d_per,
                                      unsigned char *__restrict__ dirty_next) {
  const int nl = res0 << l;
  const size_t nvox = cube((size_t)nl);
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nvox) return;
  const int kp = blockIdx.y;
  if (evaluated && evaluated[kp] == 0) return;      // see mise_subdivide_lds_kernel
  unsigned char *vs = vstate + (size_t)kp * v_per + vstate_offset(res0, l);
  if (vs[e] != 1) return;  // not a leaf (mise.pyx:236,245)
  const int s = 1 << (depth - l);
  const int vk = (int)(e % nl), vj = (int)((e / nl) % nl), vi = (int)(e / ((size_t)nl * nl));
  const int x0 = vi * s, y0 = vj * s, z0 = vk * s;
  unsigned char *ps = pstate + (size_t)kp * n_per;
  const float *vals = values + (size_t)kp * n_per;
  bool pos = false, neg = false;
  for (int a = 0; a <= s; ++a)
    for (int b = 0; b <= s; ++b)
      for (int c = 0; c <= s; ++c) {
        const size_t p = ((size_t)(x0 + a) * R1 + (y0 + b)) * R1 + (z0 + c);
        if (ps[p] == 2) {
          const double v = (double)vals[p];
          pos = pos || (v >= thr);  // :225
          neg = neg || (v <= thr);  // :227
        }
      }
  if (!(pos && neg)) return;
  vs[e] = 2;  // subdivide_voxel (:253-283)
  if (l + 1 < depth) {
    unsigned char *vc = vstate + (size_t)kp * v_per + vstate_offset(res0, l + 1);
    const int nc = nl * 2;
    for (int a = 0; a < 2; ++a)
      for (int b = 0; b < 2; ++b)
        for (int c = 0; c < 2; ++c)
          vc[((size_t)(2 * vi + a) * nc + (2 * vj + b)) * nc + (2 * vk + c)] = 1;
    if (dirty_next) mark_children(dirty_next + (size_t)kp * d_per, res0, l, nl, vi, vj, vk);
  }
  const int h = s >> 1;
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b)
      for (int c = 0; c < 3; ++c) {
        const size_t p = ((size_t)(x0 + a * h) * R1 + (y0 + b * h)) * R1 + (z0 + c * h);
        if (ps[p] == 0) ps[p] = 1;  // only add new grid points (:281-283)
      }
}

// The same level for s = 2^(depth-l) <= 4, staged through LDS.  A workgroup owns a slab of
// 1 x SUB_TJ x SUB_TK voxels; the (s+1) x (SUB_TJ*s+1) x (SUB_TK*s+1) grid points it touches
// are read once, coalesced along z, into one flag byte each (bit0: known && v >= thr,
// bit1: known && v <= thr, bit2: known).  The thread-per-voxel version above re-reads every
// point 8..27 times with a stride of s elements between lanes (3.0 ms for 256 x 32^3 voxels).
__global__ __launch_bounds__(SUB_TJ * SUB_TK) void mise_subdivide_lds_kernel(
    int R1, int res0, int depth, int l, double thr, size_t n_per, size_t v_per, size_t total_bytes,
    const float *__restrict__ values, unsigned char *__restrict__ pstate,
    unsigned char *__restrict__ vstate, const int *__restrict__ evaluated, size_t d_per,
    const unsigned char *__restrict__ dirty_cur, unsigned char *__restrict__ dirty_next) {
  extern __shared__ unsigned char flags[];
  const int nl = res0 << l;
  const int s = 1 << (depth - l);
  const int tiles_k = (nl + SUB_TK - 1) / SUB_TK, tiles_j = (nl + SUB_TJ - 1) / SUB_TJ;
  const int tile = blockIdx.x;
  const int vk0 = (tile % tiles_k) * SUB_TK, vj0 = ((tile / tiles_k) % tiles_j) * SUB_TJ;
  const int vi = tile / (tiles_k * tiles_j);
  const int kp = blockIdx.y;
  // Nothing was evaluated for this proposal in the round that has just been decoded: its points and voxels are exactly
  // as the previous subdivision pass left them, and that pass found nothing more to split (it would have queued
  // points) -- the whole proposal is a no-op.  (The tail rounds of a deep octree touch a handful of proposals.)
  if (evaluated && evaluated[kp] == 0) return;
  // A clean slab (no point of it has become known since the previous pass, no voxel of it was created by that pass) has
  // nothing to decide (see "dirty slabs" above)
  if (dirty_cur && dirty_cur[(size_t)kp * d_per + dirty_offset(res0, l) + tile] == 0) return;
  // A slab without a single leaf voxel of this level has nothing to decide either: skip it BEFORE its grid points are
  // staged.  At 128^3 the level-1 voxels exist only around the surface (~3 % of the 64^3 lattice), yet every round
  // staged all 869 M points of the level through the byte-granular path below: 27 ms per scene.
  unsigned char *vs = vstate + (size_t)kp * v_per + vstate_offset(res0, l);
  const int tk = threadIdx.x % SUB_TK, tj = threadIdx.x / SUB_TK;
  const int vk = vk0 + tk, vj = vj0 + tj;
  const size_t e = ((size_t)vi * nl + vj) * nl + vk;
  const bool leaf = vk < nl && vj < nl && vs[e] == 1;       // mise.pyx:236,245
  if (!__syncthreads_or(leaf)) return;
  unsigned char *ps = pstate + (size_t)kp * n_per;
  const float *vals = values + (size_t)kp * n_per;
  const int PK = SUB_TK * s + 1, PJ = SUB_TJ * s + 1;
  const int x0 = vi * s, y0 = vj0 * s, z0 = vk0 * s;
  if (PK == R1) {
    // the slab spans whole z rows: the rows of one x plane are one contiguous run of the
    // arrays and of `flags`.  The state bytes are fetched as aligned 16-byte words (one per
    // thread), the values only where a point is known (byte loads of the states made this
    // phase 70 % of the kernel).
    const int plane = PJ * PK;
    const int rows_in = (R1 - y0) < PJ ? (R1 - y0) : PJ;
    const int run = rows_in * PK;
    const int chunks = (plane + 30) / 16;            // 16-byte words that can touch a plane
    const unsigned char *ps_end = pstate + total_bytes;
    for (int item = threadIdx.x; item < (s + 1) * chunks; item += SUB_TJ * SUB_TK) {
      const int a = item / chunks, ch = item - a * chunks;
      const size_t base = ((size_t)(x0 + a) * R1 + y0) * R1;
      const unsigned char *g = ps + base;
      const int off = (int)(reinterpret_cast<uintptr_t>(g) & 15);
      const unsigned char *word = g - off + ch * 16;
      const int rel0 = ch * 16 - off;                // run-relative index of the word's byte 0
      if (rel0 >= plane) continue;
      unsigned w[4] = {0u, 0u, 0u, 0u};
      if (rel0 < run) {
        if (word + 16 <= ps_end) {
          const uint4 v = *reinterpret_cast<const uint4 *>(word);
          w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
        } else {
          for (int i = 0; i < 16; ++i)
            if (word + i < ps_end) w[i >> 2] |= (unsigned)word[i] << (8 * (i & 3));
        }
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int rel = rel0 + i;
        if (rel < 0 || rel >= plane) continue;
        unsigned char f = 0;
        if (rel < run && ((w[i >> 2] >> (8 * (i & 3))) & 0xffu) == 2u) {
          const double v = (double)vals[base + rel];
          f = 4 | (v >= thr ? 1 : 0) | (v <= thr ? 2 : 0);  // mise.pyx:225,227
        }
        flags[a * plane + rel] = f;
      }
    }
  } else {
    for (int t = threadIdx.x; t < (s + 1) * PJ * PK; t += SUB_TJ * SUB_TK) {
      const int c = t % PK, b = (t / PK) % PJ, a = t / (PK * PJ);
      unsigned char f = 0;
      if (y0 + b < R1 && z0 + c < R1) {
        const size_t p = ((size_t)(x0 + a) * R1 + (y0 + b)) * R1 + (z0 + c);
        if (ps[p] == 2) {
          const double v = (double)vals[p];
          f = 4 | (v >= thr ? 1 : 0) | (v <= thr ? 2 : 0);  // mise.pyx:225,227
        }
      }
      flags[t] = f;
    }
  }
  __syncthreads();
  if (!leaf) return;
  unsigned any = 0;
  for (int a = 0; a <= s; ++a)
    for (int b = 0; b <= s; ++b)
      for (int c = 0; c <= s; ++c)
        any |= flags[(a * PJ + tj * s + b) * PK + tk * s + c];
  if ((any & 3) != 3) return;
  vs[e] = 2;  // subdivide_voxel (:253-283)
  if (l + 1 < depth) {
    unsigned char *vc = vstate + (size_t)kp * v_per + vstate_offset(res0, l + 1);
    const int nc = nl * 2;
    for (int a = 0; a < 2; ++a)
      for (int b = 0; b < 2; ++b)
        for (int c = 0; c < 2; ++c)
          vc[((size_t)(2 * vi + a) * nc + (2 * vj + b)) * nc + (2 * vk + c)] = 1;
    if (dirty_next) mark_children(dirty_next + (size_t)kp * d_per, res0, l, nl, vi, vj, vk);
  }
  const int h = s >> 1;
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b)
      for (int c = 0; c < 3; ++c) {
        const size_t p = ((size_t)(x0 + a * h) * R1 + (y0 + tj * s + b * h)) * R1 + (z0 + tk * s + c * h);
        if (ps[p] == 0) ps[p] = 1;  // only add new grid points (:281-283)
      }
}

// Forward fill along one axis (mise.pyx:142-163).  Thread per line.
// axis 0: lines indexed by (j,k) run along i, etc.  `valid` = pstate >= 2.
__global__ void mise_fill_kernel(int R1, int axis, size_t n_per, float *__restrict__ values,
                                 unsigned char *__restrict__ pstate) {
  const int line = blockIdx.x * blockDim.x + threadIdx.x;
  if (line >= R1 * R1) return;
  const int kp = blockIdx.y;
  // choose the line->(u,v) mapping so that consecutive threads touch
  // consecutive z where possible
  const int u = line / R1, v = line % R1;
  size_t stride, start;
  if (axis == 0) { stride = (size_t)R1 * R1; start = (size_t)u * R1 + v; }        // (j=u,k=v)
  else if (axis == 1) { stride = R1; start = (size_t)u * R1 * R1 + v; }           // (i=u,k=v)
  else { stride = 1; start = ((size_t)u * R1 + v) * R1; }                         // (i=u,j=v)
  float *vals = values + (size_t)kp * n_per;
  unsigned char *ps = pstate + (size_t)kp * n_per;
  bool prev_valid = false;
  float prev = 0.f;
  // eight independent loads ahead of the serial carry (the stores never alias later loads)
  for (int q0 = 0; q0 < R1; q0 += 8) {
    float v[8];
    unsigned char st[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const bool in = q0 + u < R1;
      const size_t p = start + (size_t)(q0 + u) * stride;
      st[u] = in ? ps[p] : 0;
      v[u] = in ? vals[p] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (q0 + u >= R1) break;
      const size_t p = start + (size_t)(q0 + u) * stride;
      const bool valid = st[u] >= 2;
      if (!valid && prev_valid) {
        vals[p] = prev;
        ps[p] = 3;
      }
      prev = valid ? v[u] : prev;
      prev_valid = valid || prev_valid;
    }
  }
}

// y then z fill of one x plane in LDS (the two later passes of to_dense only move data inside
// a plane): the plane is read and written once, coalesced; a thread walks one line serially in
// LDS (line strides R1 and 1 words, R1 odd: conflict-free).  States are not written back --
// nothing reads them after to_dense.
__global__ __launch_bounds__(256) void mise_fill_yz_kernel(int R1, size_t n_per,
                                                           float *__restrict__ values,
                                                           const unsigned char *__restrict__ pstate) {
  extern __shared__ float plane_v[];                       // R1*R1 floats, then R1*R1 state bytes
  const int n2 = R1 * R1;
  unsigned char *plane_s = reinterpret_cast<unsigned char *>(plane_v + n2);
  const size_t base = (size_t)blockIdx.y * n_per + (size_t)blockIdx.x * n2;
  for (int t = threadIdx.x; t < n2; t += 256) {
    plane_v[t] = values[base + t];
    plane_s[t] = pstate[base + t];
  }
  __syncthreads();
  for (int axis = 1; axis <= 2; ++axis) {
    if ((int)threadIdx.x < R1) {
      const int stride = axis == 1 ? R1 : 1;
      const int start = axis == 1 ? (int)threadIdx.x : (int)threadIdx.x * R1;
      bool prev_valid = false;
      float prev = 0.f;
      for (int q = 0; q < R1; ++q) {
        const int p = start + q * stride;
        const bool valid = plane_s[p] >= 2;
        if (!valid && prev_valid) {
          plane_v[p] = prev;
          plane_s[p] = 3;
        }
        prev = valid ? plane_v[p] : prev;
        prev_valid = valid || prev_valid;
      }
    }
    __syncthreads();
  }
  for (int t = threadIdx.x; t < n2; t += 256) values[base + t] = plane_v[t];
}

// z-axis fill with a WAVE per line (lanes = k): coalesced, and the forward fill
// is a ballot + "highest valid lane below me" shuffle instead of a serial scan.
__global__ __launch_bounds__(256) void mise_fill_z_kernel(int R1, size_t n_per,
                                                          float *__restrict__ values,
                                                          unsigned char *__restrict__ pstate) {
  const int lane = threadIdx.x & 63;
  const int line = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (line >= R1 * R1) return;                 // wave-uniform
  const int kp = blockIdx.y;
  float *vals = values + (size_t)kp * n_per + (size_t)line * R1;
  unsigned char *ps = pstate + (size_t)kp * n_per + (size_t)line * R1;
  float carry = 0.f;
  bool carry_valid = false;
  for (int k0 = 0; k0 < R1; k0 += 64) {        // R1 = 65, 129, ...: 2-3 segments
    const int k = k0 + lane;
    const bool in = k < R1;
    const float v = in ? vals[k] : 0.f;
    const bool valid = in && ps[k] >= 2;
    const unsigned long long m = __ballot(valid);
    const unsigned long long below = m & ((1ull << lane) - 1ull);
    const int src = below ? 63 - __clzll((long long)below) : 0;
    const float from = __shfl(v, src);
    if (in && !valid) {
      if (below) { vals[k] = from; ps[k] = 3; }
      else if (carry_valid) { vals[k] = carry; ps[k] = 3; }
    }
    if (m) {                                   // last valid value of this segment
      const int hi = 63 - __clzll((long long)m);
      carry = __shfl(v, hi);
      carry_valid = true;
    }
  }
}

}  // namespace

RFD_API int rfd_make_grid_points(int n, float lo, float hi, float scale, float *pts,
                                 int n_padded, void *stream) {
  if (n_padded <= 0) return 0;
  hipLaunchKernelGGL(grid_points_kernel, dim3(ceil_div(n_padded, 256)), dim3(256), 0,
                     (hipStream_t)stream, n, lo, hi, scale, pts, n_padded);
  RFD_CHECK_LAUNCH();
  return 0;
}

RFD_API size_t rfd_mise_vstate_elems(int res0, int depth) {
  const size_t v = vstate_offset(res0, depth);
  return v ? v : 1;
}

RFD_API int rfd_mise_init(int K, int res0, int depth, unsigned char *pstate,
                          unsigned char *vstate, void *stream) {
  if (K <= 0) return 0;
  const int R1 = (res0 << depth) + 1;
  const size_t n_per = cube((size_t)R1);
  hipStream_t s = (hipStream_t)stream;
  RFD_CHECK(hipMemsetAsync(pstate, 0, n_per * (size_t)K, s));
  const int L = res0 + 1;
  hipLaunchKernelGGL(mise_init_points_kernel, dim3((unsigned)((L * L * L + 255) / 256), K), dim3(256),
                     0, s, R1, 1 << depth, L, n_per, pstate);
  RFD_CHECK_LAUNCH();
  const size_t v_per = rfd_mise_vstate_elems(res0, depth);
  hipLaunchKernelGGL(mise_init_voxels_kernel, dim3((unsigned)((v_per + 255) / 256), K), dim3(256),
                     0, s, depth > 0 ? cube((size_t)res0) : 0, v_per, vstate);
  RFD_CHECK_LAUNCH();
  return 0;
}

RFD_API int rfd_mise_count(int K, int res0, int depth, const unsigned char *pstate,
                           int *counts, void *stream) {
  if (K <= 0) return 0;
  const int R1 = (res0 << depth) + 1;
  const size_t n_per = cube((size_t)R1);
  hipStream_t s = (hipStream_t)stream;
  RFD_CHECK(hipMemsetAsync(counts, 0, sizeof(int) * K, s));
  hipLaunchKernelGGL(mise_count_kernel, dim3(COUNT_PARTS, K), dim3(256), 0, s, n_per, pstate, counts);
  RFD_CHECK_LAUNCH();
  return 0;
}

RFD_API int rfd_mise_collect(int K, int res0, int depth, const unsigned char *pstate,
                             const int *offsets, int *cursors, float box_size,
                             float *pts, int *lin, void *stream) {
  if (K <= 0) return 0;
  const int R1 = (res0 << depth) + 1;
  const size_t n_per = cube((size_t)R1);
  const size_t total = n_per * (size_t)K, chunks = (total + 15) / 16;
  hipLaunchKernelGGL(mise_collect_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, R1, total, n_per, K, pstate, offsets, cursors, box_size,
                     pts, lin);
  RFD_CHECK_LAUNCH();
  return 0;
}

RFD_API int rfd_mise_scatter(int n_tiles, int res0, int depth, const int *tile_prop,
                             const int *tile_src, const int *lin, const float *logits,
                             float *values, unsigned char *pstate, void *stream) {
  if (n_tiles <= 0) return 0;
  const int R1 = (res0 << depth) + 1;
  hipLaunchKernelGGL(mise_scatter_kernel, dim3(n_tiles), dim3(RFD_OCC_TILE), 0,
                     (hipStream_t)stream, cube((size_t)R1), tile_prop, tile_src, lin, logits, values,
                     pstate);
  RFD_CHECK_LAUNCH();
  return 0;
}

static int mise_subdivide(int K, int res0, int depth, double threshold, const float *values,
                          unsigned char *pstate, unsigned char *vstate, const int *evaluated, void *stream,
                          const unsigned char *dirty_cur = nullptr, unsigned char *dirty_next = nullptr) {
  if (K <= 0 || depth <= 0) return 0;
  const int R1 = (res0 << depth) + 1;
  const size_t n_per = cube((size_t)R1);
  const size_t v_per = rfd_mise_vstate_elems(res0, depth);
  const size_t d_per = dirty_offset(res0, depth);
  // fine -> coarse: children created at level l+1 were already visited this
  // round, so they cannot split until the next update (mise.pyx:239-251)
  for (int l = depth - 1; l >= 0; --l) {
    const size_t nvox = cube((size_t)res0 << l);
    const int sl = 1 << (depth - l), nl = res0 << l;
    if (sl <= 4) {
      const unsigned tiles = (unsigned)(ceil_div(nl, SUB_TK) * ceil_div(nl, SUB_TJ) * nl);
      const size_t lds = (size_t)(sl + 1) * (SUB_TJ * sl + 1) * (SUB_TK * sl + 1);
      hipLaunchKernelGGL(mise_subdivide_lds_kernel, dim3(tiles, K), dim3(SUB_TJ * SUB_TK), lds,
                         (hipStream_t)stream, R1, res0, depth, l, threshold, n_per, v_per,
                         n_per * (size_t)K, values, pstate, vstate, evaluated, d_per, dirty_cur, dirty_next);
      RFD_CHECK_LAUNCH();
      continue;
    }
    hipLaunchKernelGGL(mise_subdivide_kernel, dim3((unsigned)((nvox + 255) / 256), K), dim3(256), 0,
                       (hipStream_t)stream, R1, res0, depth, l, threshold, n_per, v_per, values,
                       pstate, vstate, evaluated, d_per, dirty_next);
    RFD_CHECK_LAUNCH();
  }
  return 0;
}

RFD_API int rfd_mise_subdivide(int K, int res0, int depth, double threshold,
                               const float *values, unsigned char *pstate,
                               unsigned char *vstate, void *stream) {
  return mise_subdivide(K, res0, depth, threshold, values, pstate, vstate, nullptr, stream);
}

// The same pass told how many points of each proposal the round just decoded has evaluated (`evaluated` [K] on the
// device: the counts rfd_mise_count produced for that round): a proposal with none is skipped -- nothing about it
// has changed since the previous pass.  Identical results; the caller passes NULL for round 0.
RFD_API int rfd_mise_subdivide_active(int K, int res0, int depth, double threshold,
                                      const float *values, unsigned char *pstate,
                                      unsigned char *vstate, const int *evaluated, void *stream) {
  return mise_subdivide(K, res0, depth, threshold, values, pstate, vstate, evaluated, stream);
}

// Bytes of one proposal's dirty-slab map (see "dirty slabs" in the kernels' part of this file).
RFD_API size_t rfd_mise_dirty_elems(int res0, int depth) { return dirty_offset(res0, depth); }

// rfd_mise_subdivide_active with dirty-slab bookkeeping.  dirty_cur / dirty_next: two [K][rfd_mise_dirty_elems] byte maps,
// zero before the first round, SWAPPED by the caller after every call.  Every call records the slabs of the voxels it
// creates in dirty_next and leaves dirty_cur zeroed.  use_dirty != 0 (a round that evaluated few points): the n_slots query
// slots of the round just decoded (lin[slot] = lattice index or < 0 for padding, tile_prop[slot / 128] = proposal) are marked
// into dirty_cur first and the pass skips every clean slab; use_dirty == 0: the pass examines every slab as
// rfd_mise_subdivide_active does (lin / tile_prop are not read).  Identical results either way.
RFD_API int rfd_mise_subdivide_dirty(int K, int res0, int depth, double threshold, const float *values,
                                     unsigned char *pstate, unsigned char *vstate, const int *evaluated,
                                     long long n_slots, const int *lin, const int *tile_prop, unsigned char *dirty_cur,
                                     unsigned char *dirty_next, int use_dirty, void *stream) {
  if (K <= 0 || depth <= 0) return 0;
  if (!dirty_cur || !dirty_next || (use_dirty && (!lin || !tile_prop || n_slots < 0))) {
    rfd_set_error("rfd_mise_subdivide_dirty: dirty maps / query list", hipErrorInvalidValue);
    return (int)hipErrorInvalidValue;
  }
  const int R1 = (res0 << depth) + 1;
  const size_t d_per = dirty_offset(res0, depth);
  if (use_dirty && n_slots > 0) {
    hipLaunchKernelGGL(mise_mark_dirty_kernel, dim3((unsigned)((n_slots + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (size_t)n_slots, R1, res0, depth, lin, tile_prop, d_per, dirty_cur);
    RFD_CHECK_LAUNCH();
  }
  const int rc = mise_subdivide(K, res0, depth, threshold, values, pstate, vstate, evaluated, stream,
                                use_dirty ? dirty_cur : nullptr, dirty_next);
  if (rc) return rc;
  RFD_CHECK(hipMemsetAsync(dirty_cur, 0, (size_t)K * d_per, (hipStream_t)stream));
  return 0;
}

RFD_API int rfd_mise_to_dense(int K, int res0, int depth, float *values,
                              unsigned char *pstate, void *stream) {
  if (K <= 0) return 0;
  const int R1 = (res0 << depth) + 1;
  const size_t n_per = cube((size_t)R1);
  const size_t plane_lds = (size_t)R1 * R1 * 5;
  const bool fused = plane_lds <= 64 * 1024 && R1 <= 256;   // a whole x plane fits in LDS
  for (int axis = 0; axis < (fused ? 1 : 2); ++axis) {
    hipLaunchKernelGGL(mise_fill_kernel, dim3(ceil_div(R1 * R1, 256), K), dim3(256), 0,
                       (hipStream_t)stream, R1, axis, n_per, values, pstate);
    RFD_CHECK_LAUNCH();
  }
  if (fused)
    hipLaunchKernelGGL(mise_fill_yz_kernel, dim3(R1, K), dim3(256), plane_lds, (hipStream_t)stream,
                       R1, n_per, values, pstate);
  else
    hipLaunchKernelGGL(mise_fill_z_kernel, dim3(ceil_div(R1 * R1, 4), K), dim3(256), 0,
                       (hipStream_t)stream, R1, n_per, values, pstate);
  RFD_CHECK_LAUNCH();
  return 0;
}
