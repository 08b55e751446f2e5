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
 *  tile_prop  [n_tiles] int32: proposal index of each tile
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
                   const void *packed, const float *fc_p_w, const float *table,
                   const float *fc_out_w, float fc_out_b, float *logits,
                   int mode, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* RFD_OCC_H */
