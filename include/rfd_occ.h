/*
 * rfd_occ.h -- C ABI of the fused occupancy decoder + batched generator
 * kernels of librfd_hip.so (MI355X / gfx950).
 *
 * Reference interface replaced (paths relative to the reference checkout):
 *   models/iscnet/modules/occ_decoder.py:110-123  DecoderCBatchNorm.forward
 *   models/iscnet/modules/layers.py:98-107        CResnetBlockConv1d.forward
 *   models/iscnet/modules/layers.py:226-242       CBatchNorm1d.forward
 *   models/iscnet/modules/occupancy_net.py:147-156 ONet.decode  (call site
 *       generator.py:137 Generator3D.eval_points)
 *   models/iscnet/modules/generator.py:78-121     generate_from_latent
 *   external/libmise/mise.pyx:87-163              MISE update / query / to_dense
 *   external/common.py:157-176                    make_3d_grid
 *
 * The reference evaluates the decoder one proposal at a time, <=100k points a
 * call, 11 CBN layers + 12 convolutions each a separate launch that round-trips
 * a (256 x T) fp32 activation through HBM.  Here the whole decoder is ONE
 * kernel: a wave owns 32 query points and all 256 channels, the residual stream
 * lives in MFMA accumulators, activations go from accumulator layout to the
 * next MFMA's B operand in registers (the weight K-order is pre-permuted to the
 * accumulator layout), weights stream from L2, and the only HBM traffic is
 * 12 B/point in + 4 B/point out.
 *
 * All pointers are device pointers unless stated.  Every function returns 0 or
 * a hipError_t value (rfd_last_error_string() in rfd_pointnet2.h).  `stream`
 * is a hipStream_t (NULL = default stream).  Nothing synchronises.
 */
#ifndef RFD_OCC_H
#define RFD_OCC_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RFD_OCC_HIDDEN 256      /* hidden_size of DecoderCBatchNorm            */
#define RFD_OCC_BLOCKS 5        /* n_blocks                                    */
#define RFD_OCC_TILE 128        /* query points per workgroup tile             */
#define RFD_OCC_TABLE_ROWS 23   /* per-proposal rows of 256 floats, see fold   */

/* Arithmetic modes of rfd_occ_decode.
 *  F16X3: every 256x256 GEMM runs as three f16 MFMAs on (hi, lo) splits of
 *         both operands (hi*hi + hi*lo + lo*hi, fp32 accumulate): ~2^-20
 *         relative error per product, i.e. fp32-class logits (the parity mode,
 *         meets the 1e-4 logit tolerance with > 10x margin).
 *  F16X1: single f16 MFMA per product (throughput mode, ~1e-3 logit error). */
#define RFD_OCC_MODE_F16X3 3
#define RFD_OCC_MODE_F16X1 1

/* Bytes of the packed weight stream written by rfd_occ_pack_weights. */
size_t rfd_occ_packed_bytes(void);

/* Re-lay the ten 256x256 fp32 weight matrices (blocks.{i}.fc_0.weight /
 * fc_1.weight, (256,256,1) Conv1d kernels, occ_decoder.py:94-96) into the MFMA
 * A-fragment stream the decode kernel consumes, in consumption order, split
 * into f16 (hi, lo) pairs after scaling by 2^kw0[i] (fc_0 of block i) or
 * 2^kw1 (all fc_1).  fc0_w / fc1_w: [5][256][256] fp32 (out, in).
 * kw0 is a HOST array of 5 ints. */
int rfd_occ_pack_weights(const float *fc0_w, const float *fc1_w,
                         const int *kw0, int kw1, void *packed, void *stream);

/* Decode n_tiles tiles of RFD_OCC_TILE query points.
 *  pts        [n_tiles*128][3] fp32 query points (pad the tail of a
 *             proposal's last tile with anything; those logits are garbage)
 *  tile_prop  [n_tiles] int32: proposal index of each tile (< 0: skip tile)
 *  tile_src   [n_tiles] int32 or NULL: tile t reads its 128 points from
 *             pts[tile_src[t]*128 ...] (NULL = identity); lets all proposals
 *             share one copy of a dense grid
 *  packed     weight stream from rfd_occ_pack_weights
 *  fc_p_w     [256][3] fp32 = fc_p.weight * 2^KH   (KH = ka + kw1)
 *  table      [K][23][256] fp32 per-proposal folded table (rfd_occ_fold.py /
 *             DESIGN.md "decoder folding"): row 0 = (fc_p.b + fc_z(z)) * 2^KH,
 *             rows 1+4i..4+4i = S0', T0', S1', T1' of block i, rows 21,22 =
 *             Sf', Tf'
 *  fc_out_w   [256] fp32, fc_out_b scalar
 *  logits     [n_tiles*128] fp32 out
 *  mode       RFD_OCC_MODE_* */
int rfd_occ_decode(int n_tiles, const float *pts, const int *tile_prop,
                   const int *tile_src, const void *packed, const float *fc_p_w, const float *table,
                   const float *fc_out_w, float fc_out_b, float *logits,
                   int mode, void *stream);

/* ---- query-point generators ---------------------------------------------------
 * Dense path (generator.py:91-97): pts[n^3][3] = scale * make_3d_grid(lo, hi, n)
 * (external/common.py:157-176: torch.linspace inclusive of both ends, x-major
 * flattening; linspace evaluated as start + step*i for i < n/2 and
 * end - step*(n-1-i) otherwise, the torch GPU kernel's formula).  The array is
 * padded with zeros up to a multiple of RFD_OCC_TILE points (n_padded). */
int rfd_make_grid_points(int n, float lo, float hi, float scale, float *pts,
                         int n_padded, void *stream);

/* ---- batched MISE (external/libmise/mise.pyx:33-369) ----------------------------
 * The reference keeps an octree (vector<Voxel>) and a hashed point list per
 * proposal on the CPU and is driven one proposal at a time.  Here the state of
 * ALL K proposals is dense and device-resident, and one launch advances every
 * proposal by one round:
 *   values [K][(R+1)^3] f32     grid values (R = resolution_0 << depth)
 *   pstate [K][(R+1)^3] u8      0 no grid point | 1 exists, unknown | 2 known
 *                               | 3 filled by to_dense
 *   vstate [K][sum_{l<depth} (resolution_0 << l)^3] u8
 *                               0 absent | 1 leaf | 2 subdivided
 * Query order is irrelevant to the result (values are scattered back by
 * lattice index), so points are handed out in whatever order the atomics give.
 */
size_t rfd_mise_vstate_elems(int res0, int depth);          /* per proposal */
int rfd_mise_init(int K, int res0, int depth, unsigned char *pstate,
                  unsigned char *vstate, void *stream);
/* counts[k] = number of existing-but-unknown grid points (mise.pyx:111-114) */
int rfd_mise_count(int K, int res0, int depth, const unsigned char *pstate,
                   int *counts, void *stream);
/* Write each proposal's unknown points into its tile-aligned segment:
 * offsets[k] = first point slot of proposal k (multiple of 128), cursors[k]
 * zero-initialised scratch.  pts[slot] = box_size * (c / R - 0.5)
 * (generator.py:106-109), lin[slot] = lattice index.  Padding slots must be
 * pre-set by the caller (lin = -1). */
int rfd_mise_collect(int K, int res0, int depth, const unsigned char *pstate,
                     const int *offsets, int *cursors, float box_size,
                     float *pts, int *lin, void *stream);
/* values[prop][lin] = logits[slot], pstate = known (mise.pyx:96-102).  tile_src (optional):
 * tile of `lin` holding this tile's lattice indices, for query lists shared by several
 * proposals (round 0: every proposal asks for the same level-0 lattice). */
int rfd_mise_scatter(int n_tiles, int res0, int depth, const int *tile_prop,
                     const int *tile_src, const int *lin, const float *logits,
                     float *values, unsigned char *pstate, void *stream);
/* One subdivide_voxels pass (mise.pyx:196-251): a leaf voxel below max depth
 * splits iff among the known points of its closed cube one has value >= thr
 * and one has value <= thr (both non-strict, :225-227). */
int rfd_mise_subdivide(int K, int res0, int depth, double threshold,
                       const float *values, unsigned char *pstate,
                       unsigned char *vstate, void *stream);
/* The same pass, told how many points of each proposal the round just decoded has evaluated (`evaluated` [K], device:
 * the counts rfd_mise_count produced for that round; NULL = every proposal, as above).  A proposal with none is left
 * alone -- the reference stops calling update() / subdivide_voxels() for an object as soon as its query() is empty
 * (generator.py:104-117), so its octree never changes again.  Same results, and the tail rounds of a deep octree touch
 * a handful of proposals instead of all K lattices. */
int rfd_mise_subdivide_active(int K, int res0, int depth, double threshold, const float *values,
                              unsigned char *pstate, unsigned char *vstate, const int *evaluated,
                              void *stream);
/* The same pass with dirty-slab bookkeeping, for octrees whose last rounds evaluate a few thousand points (generator.py:99-117:
 * the loop runs until query() is empty, and every pass of the plain entry stages every slab that holds a leaf voxel).  A leaf
 * can only split if a point of its closed cube has become known since the previous pass or if that pass created it, so:
 * dirty_cur / dirty_next are two [K][rfd_mise_dirty_elems(res0, depth)] byte maps, zero before the first round and SWAPPED
 * by the caller after every call; every call records the slabs of the voxels it creates in dirty_next and leaves dirty_cur
 * zeroed.  use_dirty != 0: the n_slots query slots of the round just decoded (lin[slot] = lattice index, < 0 = padding;
 * tile_prop[slot / 128] = proposal) are marked into dirty_cur first and clean slabs are skipped; use_dirty == 0: every slab
 * is examined (lin / tile_prop are not read).  Results identical to rfd_mise_subdivide_active either way. */
size_t rfd_mise_dirty_elems(int res0, int depth);           /* per proposal */
int rfd_mise_subdivide_dirty(int K, int res0, int depth, double threshold, const float *values,
                             unsigned char *pstate, unsigned char *vstate, const int *evaluated, long long n_slots,
                             const int *lin, const int *tile_prop, unsigned char *dirty_cur,
                             unsigned char *dirty_next, int use_dirty, void *stream);
/* to_dense (mise.pyx:133-163): forward-fill along x, then y, then z.  pstate is working
 * storage here: its content after the call is unspecified (only `values` is the result). */
int rfd_mise_to_dense(int K, int res0, int depth, float *values,
                      unsigned char *pstate, void *stream);

/* ---- batched marching cubes (generator.py:157-161; PyMCubes 0.1.2 is not
 * vendored: published algorithm restated, ordering is ours) ----------------------
 * grids [K][n][n][n] f32; the -1e6 padding shell of generator.py:158-159 is
 * virtual: D = n + 2 lattice points per axis.  Workgroups own rfd_mc_blocks(n)
 * runs of consecutive lattice points per proposal.  classify fills code[K][D^3]
 * (the 8-bit cube index of the cell whose origin is the lattice point; the crossed
 * +x/+y/+z edges = vertices the point owns, and the triangle count, follow from
 * it) and the per-workgroup totals vsum / tsum
 * [K][rfd_mc_blocks(n)].  The caller takes EXCLUSIVE prefix sums vblock / tblock of
 * the flattened totals, sizes the outputs, then emit writes verts [NV][3] f64 in
 * padded-grid index coordinates (original grid point i sits at i + 1) and tris
 * [NT][3] i32 with vertex indices local to each proposal; vbase [K][D^3] i32 is
 * scratch (no initialisation needed).  Corner c is "outside" when value < iso;
 * triangle normals point outside. */
int rfd_mc_blocks(int n);
int rfd_mc_classify(int K, int n, float pad_value, double iso, const float *grids,
                    unsigned char *code, int *vsum, int *tsum, void *stream);
int rfd_mc_emit(int K, int n, float pad_value, double iso, const float *grids,
                const unsigned char *code, const int *vblock, const int *tblock,
                int *vbase, double *verts, int *tris, void *stream);
/* the same, storing va * vertex + vc (one fma per coordinate): Generator3D.extract_mesh's `v -= 0.5; v -= 1;
 * v /= n - 1; v = box * (v - 0.5)` (generator.py:163-168) is such a map, applied here instead of in a second pass */
int rfd_mc_emit_affine(int K, int n, float pad_value, double iso, const float *grids,
                       const unsigned char *code, const int *vblock, const int *tblock,
                       int *vbase, double *verts, int *tris, double va, double vc, void *stream);

/* ---- proposal post-processing (net_utils/ap_helper.py:131-264 parse_predictions,
 * net_utils/nms.py:79-118, net_utils/libs.py:128-137: CPU numpy + scipy Delaunay
 * in the reference) ---------------------------------------------------------------
 * boxes  [b][K][7] f64: centre xyz, size (l,w,h), heading angle, in the scan frame;
 * pts    [b][n][point_stride] f32 (xyz first); counts [b][K] = points inside each
 * oriented box (the reference keeps boxes with >= 5, ap_helper.py:196). */
int rfd_points_in_boxes(int b, int K, int n, int point_stride, const float *pts,
                        const double *boxes, int *counts, void *stream);
/* Greedy NMS on axis-aligned boxes aabb [b][K][6] f64 (x1,y1,z1,x2,y2,z2);
 * order [b][K] i32 = box indices by descending score (sorted by the caller);
 * `valid` selects the participating boxes, `keep` receives the picks.  use_cls: only boxes of the same class suppress each other
 * (nms_3d_faster_samecls); old_type: overlap / area of the other box instead of
 * IoU.  K <= 1024. */
int rfd_nms3d(int b, int K, double iou_thr, int old_type, int use_cls, const double *aabb,
              const int *order, const int *cls, const unsigned char *valid,
              unsigned char *keep, void *stream);

/* Eight-wave variant of the fused decoder (csrc/occ_decoder8.hip): same arguments, results and
 * table as rfd_occ_decode; its weight stream has a different fragment order (16x16x32 MFMA) and
 * is produced by rfd_occ_pack_weights_w8 into a buffer of rfd_occ_packed_bytes() bytes. */
int rfd_occ_pack_weights_w8(const float *fc0_w, const float *fc1_w, const int *kw0, int kw1,
                            void *packed, void *stream);
int rfd_occ_decode_w8(int n_tiles, const float *pts, const int *tile_prop, const int *tile_src,
                      const void *packed, const float *fc_p_w, const float *table,
                      const float *fc_out_w, float fc_out_b, float *logits, int mode, void *stream);
/* The same launch with MISE.update (mise.pyx:87-104, the value / known part) fused into its epilogue:
 * instead of a tile-ordered logits buffer, query slot s writes values[prop][lin[s]] (prop = tile_prop of
 * its tile, values / pstate [K][n_per]) and sets pstate[prop][lin[s]] = 2 (known); lin[s] < 0 = padding
 * slot.  lin is indexed like pts (through tile_src when given).  Replaces rfd_occ_decode_w8 +
 * rfd_mise_scatter (generator.py:110-117: decode, then mesh_extractor.update). */
int rfd_occ_decode_scatter_w8(int n_tiles, const float *pts, const int *tile_prop, const int *tile_src,
                              const void *packed, const float *fc_p_w, const float *table,
                              const float *fc_out_w, float fc_out_b, const int *lin, float *values,
                              unsigned char *pstate, long long n_per, int mode, void *stream);

/* The eight-wave decoder hands its tiles out at run time (a persistent workgroup whose CU is still busy with another
 * stream's waves at launch starts late; with a static partition it would set the kernel's end): chunk k of a launch
 * of n_tiles tiles on n_workgroups workgroups covers tiles [*begin, *end) -- batches of n_workgroups equal chunks,
 * each batch half of what is left, single tiles at the end; *begin == *end == n_tiles past the last chunk.  Host-side
 * view of the schedule for tests and tools; the kernel evaluates the same function. */
int rfd_occ_chunk_range(int k, int n_tiles, int n_workgroups, int *begin, int *end);
/* Launch shape of the eight-wave decoder; every argument >= 0 sets, < 0 leaves unchanged.  The default (0, 0, 0) =
 * persistent grid of one workgroup per CU with run-time chunk claiming is what ships; the others are the A/B controls
 * of the tests and of profiles/r04_*.txt, bit-identical in their results and slower: static_partition != 0 = rounds 1-3's
 * equal runs of consecutive tiles per workgroup; cus = n: a persistent grid of n < num_cu workgroups; chunk_cap = c
 * (1..255): NOT persistent, one workgroup per chunk of at most c tiles.  Initial values come ONCE from the environment
 * (RFD_DECODER_STATIC, RFD_DECODER_CUS, RFD_DECODER_CHUNK); nothing reads the environment on the launch path. */
int rfd_occ_set_launch_shape(int static_partition, int cus, int chunk_cap);
/* Tail launches (csrc/occ_decoder_tail.hip): a decode of at most n_tiles tiles (default 384; RFD_DECODER_TAIL_TILES
 * once at start-up) runs on a one-wave, no-LDS kernel that can start on a CU other kernels are using, instead of waiting
 * for empty CUs -- the last MISE rounds of generator.py:99-117's loop, a few thousand points per scene.  Logits bit-identical
 * to the main kernel's; 16-slot groups that are all padding are skipped.  n_tiles = 0: never; < 0: leave unchanged.
 * Returns the previous setting. */
int rfd_occ_set_tail_tiles(int n_tiles);
/* The same schedule with no chunk larger than max_chunk tiles (max_chunk = 0: uncapped) -- the shape of the
 * one-workgroup-per-chunk launch (rfd_occ_set_launch_shape's chunk_cap); *n_chunks (optional) = number of non-empty chunks = its grid. */
int rfd_occ_chunk_range_capped(int k, int n_tiles, int n_workgroups, int max_chunk, int *begin, int *end,
                               int *n_chunks);

/* ---- fp32-class GEMM on the f16 matrix cores (csrc/gemm_f16x3.hip) -----------------
 * C[M,N] = act(A)[M,K] . W[N,K]^T (+ bias[N]) (+ gbias[m / rows_per_group][N]) (+ R[M,N]),
 * optional ReLU on A and on C; three f16 MFMAs per product on (hi, lo) operand splits.
 * Replaces the fp32 GEMMs of the skip-propagation encoder (layers.py:340-392).
 * W is re-laid once with rfd_gemm_pack_w (scaled by 2^sw); A is scaled by 2^sa on the fly.
 * M % 128 == N % 128 == K % 32 == 0, lda % 4 == 0.
 * pool_max (optional, [M / rows_per_group][N], zero-initialised by the caller): running
 * max(0, C) over the rows of each group = the encoder's max-pool + ReLU (layers.py:380-392)
 * fused into the epilogue; needs the row-owner kernel (M, N % 256, K % 128, rows_per_group % 64).
 * With pool_max, C may be NULL: the product is then only pooled, never written.
 * pool_signed != 0: pool_max receives the plain max over the rows instead (the caller
 * initialises it to -inf), for consumers that do not rectify (pointseg.py global feature). */
size_t rfd_gemm_packed_bytes(int N, int K);
int rfd_gemm_pack_w(int N, int K, int sw, const float *W, void *packed, void *stream);
int rfd_gemm_f16x3(int M, int N, int K, const float *A, int lda, const void *packed_w,
                   float *C, int ldc, const float *bias, const float *gbias,
                   int rows_per_group, const float *R, int ldr, int relu_in, int relu_out,
                   int sa, int sw, float *pool_max, int pool_signed, void *stream);

/* ---- small elementwise passes (csrc/small_ops.hip) -------------------------------------------------------------------
 * rfd_occ_fold_rows: DecoderCBatchNorm's per-proposal table (layers.py:226-242 folded: rfdnet_amd/occ_fold.py) from the
 * stacked gamma / beta products gb [K][2L][H]: scale = gb[k][l] / sqrtv[l], shift = gb[k][L+l] - mean[l] scale,
 * table[k][1+2l] = scale smul[l], table[k][2+2l] = (shift + scale extra[l]) tmul[l], table[k][0] = row0[k] (row0_stride
 * = H, or 0 for one shared row); every operation rounded on its own (bit-identical to the torch composition).
 * rfd_rows3_rotate_z / rfd_rows3_affine: STN_Group's two point transforms on rows [G][P][3] (pointnet2_modules.py:517-527,
 * :452-466): rotation by the group's (cos, sin) about z; the learned 3 x 4 affine A [G][3][4]. */
int rfd_occ_fold_rows(int K, int L, int H, const float *gb, const float *sqrtv, const float *mean, const float *extra,
                      const float *smul, const float *tmul, const float *row0, int row0_stride, float *table, void *stream);
int rfd_rows3_rotate_z(int G, int P, const float *rows, const float *cs, float *out, void *stream);
int rfd_rows3_affine(int G, int P, const float *rows, const float *A, float *out, void *stream);

/* ---- fragment-ordered split activations ("frag rows", csrc/gemm_f16x3.hip) --------------------------------------
 * Every consumer of an encoder activation rectifies it (layers.py:27,38-46: the in-place ReLU), so a producer can
 * store relu(x) 2^sa ONCE, already split into f16 (hi, lo) -- the same 4 bytes per element as fp32 -- in the operand
 * order of the consumer's matrix instruction.  Layout of M rows x C channels (M % 32 == 0, C % 32 == 0):
 *   [M/32 row blocks][C/32 channel blocks][k step 2][hi, lo][lane 64][8 f16]      (4 KiB per 32 x 32 block)
 *   row = 32 rb + (lane & 31);  channel = 32 kb + (r & 3) + 8 (r >> 2) + 4 (lane >> 5),  r = 8 kstep + j
 * A buffer is addressed by the pointer to its first block and the byte stride between row blocks; a channel window is
 * a pointer offset of 4096 bytes per channel block.  `sa` is a property of the data: producer and consumer agree on it.
 * rfd_rows_to_frag / rfd_frag_to_rows convert from / to fp32 rows ((hi + lo) 2^-sa is exact).
 * rfd_gemm_f16x3_frag: C_frag = split(relu(A W^T + bias + gbias[m / rows_per_group]) 2^sa), A given as frag rows;
 * M % 256 == 0, N % 256 == 0, K % 128 == 0, rows_per_group % 64 == 0 when gbias / pool_max is given (gbias_stride = floats
 * between its rows, 0 = N: gbias may be a column window of a wider per-group matrix); packed_w from
 * rfd_gemm_pack_w.  C_frag may be NULL with pool_max ([M / rows_per_group][N]: max over the group's rows of the fp32
 * result -- max(0, .) into a zero-initialised pool, the plain max into a -inf-initialised one with pool_signed).
 * Replaces, per ResnetBlockFC of the encoder (layers.py:5-48, 340-392), the ReLU + scale + split that every GEMM
 * re-did on its fp32 input, and the epilogue's transposition through LDS. */
size_t rfd_frag_bytes(int M, int C);
int rfd_rows_to_frag(int M, int C, const float *x, int ldx, int relu, int sa, void *out, long rb_stride, void *stream);
int rfd_frag_to_rows(int M, int C, const void *in, long rb_stride, int sa, float *x, int ldx, void *stream);
int rfd_gemm_f16x3_frag(int M, int N, int K, const void *A_frag, long a_rb_stride, const void *packed_w,
                        void *C_frag, long c_rb_stride, const float *bias, const float *gbias, int gbias_stride,
                        int rows_per_group, int sa, int sw, float *pool_max, int pool_signed, void *stream);

/* ---- first layer of the skip-propagation point encoder (csrc/pos_embed.hip) ------------
 * fc_pos applied to cat([points, box feature]) * mask (skip_propagation.py:55-66,
 * layers.py:364-366), with the per-proposal share hoisted out by the caller:
 *   out[r][n] = bias[n] + mask[r] * (sum_{j<d} x[r][j] W[n][j] + group[r / rows_per_group][n])
 * x [M][ldx] (first d <= 8 columns), mask [M], W [N][ldw] (first d columns), bias [N],
 * group [M / rows_per_group][N] = box_feature . W[:, d:]^T, out [M][ldo] (may be a column
 * window of a wider row-major buffer).  N % 4 == 0, ldo % 4 == 0.  sa = the activation scale exponent of
 * the split-precision GEMM that consumes `out` (status bit 4 when |out| 2^sa would leave the f16 range). */
int rfd_pos_embed(int M, int N, int d, const float *x, int ldx, const float *mask,
                  const float *W, int ldw, const float *bias, const float *group,
                  int rows_per_group, float *out, int ldo, int sa, void *stream);
/* the same layer written as frag rows (above) of relu(out) 2^sa: out = first block, rb_stride = bytes between row
 * blocks; M % 32 == 0, N % 32 == 0, rows_per_group % 32 == 0, 1 <= d <= 8 */
int rfd_pos_embed_frag(int M, int N, int d, const float *x, int ldx, const float *mask,
                       const float *W, int ldw, const float *bias, const float *group,
                       int rows_per_group, void *out, long rb_stride, int sa, void *stream);

/* ---- PointNet feature chains of the skip-propagation nets, fused (csrc/pointseg_chain.hip) ----------------
 * models/iscnet/modules/pointseg.py:7-42 (STN3d), :45-79 (STNkd), :82-129 (PointNetEncoder): per point
 *   [d_in -> 64, ReLU] -> 64 -> 128, ReLU -> 128 -> 1024 [, ReLU] -> max over the P points of a proposal
 * with the BatchNorms folded into W / b by the caller.  mode 1: first layer on d_in <= 8 input columns (STN3d);
 * mode 2: first layer 64 -> 64 (STNkd); mode 0: no first layer, x is the 64-wide point feature (encoder conv2/3).
 * rfd_chain_pack splits the weights (scaled by 2^sw, |w| 2^sw <= 2^14) into f16 (hi, lo) MFMA fragments:
 * rfd_chain_packed_bytes() bytes.  rfd_chain_pool: x [M][ldx] fp32 rows, P % 512 == 0, M % P == 0,
 * out [M / P][1024].  sa = activation scale exponent (status bit 4 when |activation| 2^sa leaves the f16 range). */
size_t rfd_chain_packed_bytes(void);
int rfd_chain_pack(int mode, const float *W1, const float *W2, const float *W3, int sw1, int sw2, int sw3,
                   void *packed, void *stream);
int rfd_chain_pool(int mode, int M, int P, int d_in, const float *x, int ldx, const void *packed,
                   const float *W1raw, const float *b1, const float *b2, const float *b3, int relu3, int sa,
                   int sw1, int sw2, int sw3, float *out, void *stream);
/* the same chain with a last layer of c3 channels, a multiple of 64 up to 1024 (W3 [c3][128], b3 [c3], out [M / P][c3]):
 * 256 = the STN3d of STN_Group (pointnet2_modules.py:420-466: conv 3 -> 64 -> 128 -> 256 + BatchNorms + ReLU + max over
 * the group's points); rfd_chain_pack / rfd_chain_pool are the c3 = 1024 case */
size_t rfd_chain_packed_bytes_n(int c3);
int rfd_chain_pack_n(int mode, int c3, const float *W1, const float *W2, const float *W3, int sw1, int sw2, int sw3,
                     void *packed, void *stream);
int rfd_chain_pool_n(int mode, int c3, int M, int P, int d_in, const float *x, int ldx, const void *packed,
                     const float *W1raw, const float *b1, const float *b2, const float *b3, int relu3, int sa,
                     int sw1, int sw2, int sw3, float *out, void *stream);

/* PointSeg's per-point head fused (csrc/pointseg_chain.hip; pointseg.py:131-154 with the BatchNorms folded):
 *   x [M][ldx] (64-wide point feature) -> 64 -> 512 (+ gbias[row / P], ReLU) -> 256 (ReLU) -> 128 (ReLU) -> n_cls scores
 * gbias [M / P][512] = conv1's bias + its global-feature columns applied to the proposal's global feature (one vector per
 * proposal, computed by the caller).  Wa [512][64], Wb [256][512], Wc [128][256] packed by rfd_head_pack
 * (rfd_head_packed_bytes() bytes); Wd [n_cls][128], bd [n_cls] fp32, n_cls <= 2; P % 128 == 0; out [M][n_cls]. */
size_t rfd_head_packed_bytes(void);
int rfd_head_pack(const float *Wa, const float *Wb, const float *Wc, int swa, int swb, int swc, void *packed,
                  void *stream);
int rfd_head_scores(int M, int P, const float *x, int ldx, const void *packed, const float *gbias, const float *bb,
                    const float *bc, const float *Wd, const float *bd, int n_cls, int sa, int swa, int swb, int swc,
                    float *out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* RFD_OCC_H */
