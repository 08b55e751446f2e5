/*
 * rfd_pointnet2.h -- C ABI of the MI355X (gfx950) PointNet++ operator library
 * (librfd_hip.so).  Drop-in boundary for RfD-Net's `pointnet2_ops._ext`.
 *
 * Every entry point below replaces one `*_kernel_wrapper` that the reference's
 * C++ host layer declares and its .cu files define:
 *
 *   reference declaration (external/pointnet2_ops_lib/pointnet2_ops/_ext-src/)
 *   ----------------------------------------------------------------------------
 *   src/sampling.cpp:4-13      gather_points[_grad]_kernel_wrapper,
 *                              furthest_point_sampling_kernel_wrapper
 *   src/ball_query.cpp:4-6     query_ball_point_kernel_wrapper
 *   src/group_points.cpp:4-10  group_points[_grad]_kernel_wrapper
 *   src/interpolate.cpp:4-12   three_nn / three_interpolate[_grad]_kernel_wrapper
 *
 * Same names, argument order and meaning; the ONE addition is a trailing
 * `void* stream` (a hipStream_t; NULL = the default stream) because the
 * reference fetches `at::cuda::getCurrentCUDAStream()` implicitly
 * (sampling_gpu.cu:26,180) and a C ABI cannot.  All launches are asynchronous
 * on that stream; nothing here synchronises or allocates per call (a small
 * per-device workspace for the multi-workgroup FPS exchange is allocated once,
 * lazily).  Pointers are device pointers on the CURRENT HIP device; layouts are
 * dense row-major exactly as in the reference (CHECK_CONTIGUOUS, utils.h:10-13).
 *
 * Error behaviour: the reference prints and exit(-1)s on a launch failure
 * (cuda_utils.h:30-39).  Here every function returns 0 on success or a
 * hipError_t value; rfd_last_error_string() describes the last failure of the
 * calling thread.  Callers raise instead of exiting.
 *
 * Differences the caller may rely on (supersets of the reference contract):
 *   - query_ball_point writes EVERY element of idx (rows without a neighbour
 *     are written as zeros), so idx need not be pre-zeroed (the reference host
 *     zero-fills it, ball_query.cpp:19-21; pre-zeroing remains harmless).
 *   - furthest_point_sampling requires `temp` (b*n floats) like the reference
 *     and leaves in it the same final min-distances the CUDA kernel leaves;
 *     it need not be pre-filled with 1e10 (sampling.cpp:74-76) -- the kernel
 *     initialises it.
 */
#ifndef RFD_POINTNET2_H
#define RFD_POINTNET2_H

#ifdef __cplusplus
extern "C" {
#endif

/* sampling.cpp:11-13 / sampling_gpu.cu:176-229.
 * dataset (b,n,3) f32, temp (b,n) f32 scratch, idxs (b,m) i32 out. */
int furthest_point_sampling_kernel_wrapper(int b, int n, int m,
                                           const float *dataset, float *temp,
                                           int *idxs, void *stream);

/* sampling.cpp:4-6 / sampling_gpu.cu:22-30.
 * points (b,c,n) f32, idx (b,npoints) i32, out (b,c,npoints) f32. */
int gather_points_kernel_wrapper(int b, int c, int n, int npoints,
                                 const float *points, const int *idx,
                                 float *out, void *stream);

/* sampling.cpp:7-9 / sampling_gpu.cu:49-57.  grad_points (b,c,n) must be
 * zero-initialised by the caller (sampling.cpp:49-51); scatter-add. */
int gather_points_grad_kernel_wrapper(int b, int c, int n, int npoints,
                                      const float *grad_out, const int *idx,
                                      float *grad_points, void *stream);

/* ball_query.cpp:4-6 / ball_query_gpu.cu:46-54.
 * new_xyz (b,m,3), xyz (b,n,3) f32, idx (b,m,nsample) i32 out. */
int query_ball_point_kernel_wrapper(int b, int n, int m, float radius,
                                    int nsample, const float *new_xyz,
                                    const float *xyz, int *idx, void *stream);

/* group_points.cpp:4-6 / group_points_gpu.cu:30-38.
 * points (b,c,n) f32, idx (b,npoints,nsample) i32, out (b,c,npoints,nsample). */
int group_points_kernel_wrapper(int b, int c, int n, int npoints, int nsample,
                                const float *points, const int *idx,
                                float *out, void *stream);

/* group_points.cpp:8-10 / group_points_gpu.cu:66-75.  grad_points (b,c,n)
 * zero-initialised by the caller (group_points.cpp:48-50). */
int group_points_grad_kernel_wrapper(int b, int c, int n, int npoints,
                                     int nsample, const float *grad_out,
                                     const int *idx, float *grad_points,
                                     void *stream);

/* interpolate.cpp:4-5 / interpolate_gpu.cu:61-68.
 * unknown (b,n,3), known (b,m,3) -> dist2 (b,n,3) SQUARED, idx (b,n,3). */
int three_nn_kernel_wrapper(int b, int n, int m, const float *unknown,
                            const float *known, float *dist2, int *idx,
                            void *stream);

/* interpolate.cpp:6-8 / interpolate_gpu.cu:103-111.
 * points (b,c,m), idx/weight (b,n,3) -> out (b,c,n). */
int three_interpolate_kernel_wrapper(int b, int c, int m, int n,
                                     const float *points, const int *idx,
                                     const float *weight, float *out,
                                     void *stream);

/* interpolate.cpp:9-12 / interpolate_gpu.cu:145-154.  grad_points (b,c,m)
 * zero-initialised by the caller (interpolate.cpp:85-87). */
int three_interpolate_grad_kernel_wrapper(int b, int c, int n, int m,
                                          const float *grad_out,
                                          const int *idx, const float *weight,
                                          float *grad_points, void *stream);

/* ---- fused forms (no reference counterpart; used by the host mirror of
 * QueryAndGroup, pointnet2_utils.py:302-361, to avoid materialising and
 * re-reading the grouped tensor).  Results are bit-identical to composing the
 * reference ops ON A GPU: group(xyz^T, idx) - centre [ * (1.0f / radius), which
 * is how torch divides a device tensor by a Python scalar ] and group(features). */

/* out (b, 3+c, m, nsample): channels 0..2 = (xyz[idx] - new_xyz) [/ radius if
 * normalize], channels 3.. = features[:, idx].  features may be NULL (c = 0).
 * If grouped_xyz_out != NULL it receives channels 0..2 as (b,3,m,nsample).
 * xyz (b,n,3), new_xyz (b,m,3), features (b,c,n), idx (b,m,nsample). */
int rfd_group_concat(int b, int c, int n, int m, int nsample, float radius,
                     int normalize, int use_xyz, const float *xyz,
                     const float *new_xyz, const float *features,
                     const int *idx, float *out, float *grouped_xyz_out,
                     void *stream);

/* FPS that also emits the sampled centres: new_xyz (b,m,3) = dataset[idxs]
 * (== gather_points(dataset^T, idxs)^T, pointnet2_modules.py:224-226). */
int rfd_furthest_point_sampling_gather(int b, int n, int m,
                                       const float *dataset, float *temp,
                                       int *idxs, float *new_xyz,
                                       void *stream);

/* ---- diagnostics --------------------------------------------------------- */
const char *rfd_last_error_string(void);
/* Device-side status words of the persistent kernels (bit 0: a multi-workgroup FPS launch
 * ABORTED on its exchange time-out, see rfd_fps_set_timeout_ms; bit 1: occupancy decoder,
 * bit 2: split-precision GEMMs -- an activation beyond the f16 range at the current scale).
 * One word per stream (64 slots per device; slot 0 is the null stream's and the overflow slot).
 * rfd_device_status synchronises the DEVICE and returns / clears the OR of all words.  0 = OK. */
int rfd_device_status(void);
/* The word of `stream` only, after waiting for that stream (other streams keep
 * running and keep their own flags); cleared when reported. */
int rfd_stream_status(void *stream);
/* Asynchronous snapshot: the word is copied to *host_word (PINNED host memory; valid after the
 * caller's next synchronisation of `stream`) and reset, in stream order.  Nothing waits. */
int rfd_stream_status_snapshot(void *stream, unsigned *host_word);
/* Give the status slot of `stream` back (callers that create a stream per scene);
 * waits for the stream and returns its pending flags.  Not owning a slot is fine. */
int rfd_release_stream(void *stream);
/* Multi-workgroup furthest point sampling (n > 4096 points per scene: the scene is spread over G <= 64 workgroups
 * that exchange their candidates every round) needs its G workgroups resident together.  Where they cannot be -- a
 * partitioned GPU, a CU-masked queue, another stream's or process's persistent kernel holding the CUs -- the launch does NOT
 * hang: a workgroup whose polling wave has waited `ms` milliseconds (wall clock; default 500, or RFD_FPS_TIMEOUT_MS at
 * load) for one round's candidates raises the launch's sticky abort word, every workgroup of the launch -- running, or
 * dispatched only later -- leaves at once, status bit 0 is raised on the stream and idxs keeps the caller's
 * zero-fill beyond the round reached.  (Measured on MI355X: beside a kernel that holds most CUs the hardware usually
 * places NONE of the grid's workgroups and the launch just waits for the CUs, as any launch would; a partially placed
 * grid -- seen with 8 free CUs on the null stream -- ends through the abort.)  The reference's answer to a launch that cannot run is to fail fast as well
 * (cuda_utils.h:30-39: message + exit); here the host raises and the device stays usable.  Returns the previous
 * value; ms <= 0 restores the default. */
int rfd_fps_set_timeout_ms(int ms);
/* Points per thread of the multi-workgroup FPS kernel (5, 8, 10, 16, 20, 32, 40, 64; 0 = the launcher's own choice,
 * 10 at 80 000 points = 32 workgroups).  Geometry sweeps (tools/fps_sweep.py) and tests; results never depend on
 * it.  Returns the previous value, -2 for a value that is not instantiated. */
int rfd_fps_set_geometry(int points_per_thread);
/* The two TEST HOOKS below are compiled out by -DRFD_NO_TEST_HOOKS (`RFD_NO_TEST_HOOKS=1 python -m rfdnet_amd.build`: a
 * deployment build; tests/test_gpu_fps_abort.py then skips).  The default build carries them so that the library the
 * GPU tests exercise is the library that ships.
 * TEST HOOK, not part of the product path: every round of the NEXT multi-workgroup FPS call (one-shot: the call
 * consumes it) also waits for `n` (0..8) exchange units that nobody publishes -- what a workgroup that is never
 * dispatched looks like to the resident ones; the deterministic way into the time-out above.  0 = off.  Returns the
 * previous value. */
int rfd_fps_test_phantom_units(int n);
/* TEST HOOK, not part of the product path: occupy all but `leave_free_cus` compute units of the current device with
 * workgroups that each hold a CU's whole LDS, until *release_flag (device memory) becomes non-zero or max_ms (<= 10000)
 * have passed -- a multi-workgroup FPS launched beside it cannot have all its workgroups resident, which is the
 * situation the time-out above exists for (tests/test_gpu_fps_abort.py).  Returns the number of holding workgroups
 * or a negative hipError. */
int rfd_test_hold_cus(int leave_free_cus, const unsigned *release_flag, int max_ms, void *stream);
/* "gfx950" etc. of the code object actually loaded. */
const char *rfd_build_arch(void);

/* Feature propagation up to its shared MLP, fused (PointnetFPModule.forward, pointnet2_modules.py:383-392 + ThreeNN's
 * sqrt, pointnet2_utils.py:124-125): out [b][c + cs][n] = cat([three_interpolate(points [b][c][m], idx, w), skip
 * [b][cs][n]], 1) with w_t = r_t / ((r_0 + r_1) + r_2), r_t = 1 / (sqrt(dist2_t) + 1e-8); dist2 / idx [b][n][3] as
 * three_nn_kernel_wrapper returns them.  Replaces seven launches (sqrt, add, reciprocal, sum, divide, interpolate, cat). */
int rfd_three_interpolate_cat(int b, int c, int cs, int m, int n, const float *points, const int *idx,
                              const float *dist2, const float *skip, float *out, void *stream);
/* A chain of 1 .. 4 pointwise layers on a channel-major tensor, one kernel (csrc/mlp_cols.hip): x [B][widths[0]][N] ->
 * y [B][widths[n_layers]][N], layer i = W_i a + b_i (+ ReLU when relu[i]); wt[i] = W_i TRANSPOSED, rows zero padded to a
 * multiple of four: [C_(i-1)][(C_i + 3) & ~3], 16-byte aligned, with an eval-mode BatchNorm folded in by the caller;
 * exact fp32 fma arithmetic (the k range of a layer is summed in up to 8 interleaved slices, then bias + slices in order).
 * N % 8 == 0, widths <= 1024.  The shared MLPs of PointnetFPModule (pointnet2_modules.py:395-403), VotingModule
 * (vote_module.py:34-61) and ProposalModule's head (proposal_module.py:85-124): conv + BatchNorm + ReLU launches. */
int rfd_mlp_cols(int B, int N, int n_layers, const int *widths, const float *const *wt, const float *const *bias,
                 const int *relu, const float *x, float *y, void *stream);

/* One set-abstraction layer after the ball query, fused: group -> centre-subtract
 * [* (1.0f / radius)] -> concat(xyz first) -> 3 x [1x1 conv + BN(eval) + ReLU] -> max over
 * the nsample neighbours, without materialising the (3+c, m, nsample) tensor.  Replaces
 * PointnetSAModuleVotes.forward after the grouper's ball query (pointnet2_modules.py:219-255,
 * pointnet2_utils.py:333-344, build_shared_mlp :9-19) at inference; exact-fp32 MFMA.
 * w1..w3: BN-folded weights re-laid by the host (rfdnet_amd/sa_fused.py documents the layout),
 * b1..b3 folded biases; out (b, c3, m).  nsample in {16, 32, 64}; instantiated widths:
 * (3+c_feat -> c1, c2, c3) = (4 -> 64,64,128), (131 -> 128,128,256), (259 -> 128,128,256),
 * (259 -> 128,128,128). */
int rfd_sa_fused(int b, int n, int m, int nsample, int c_feat, float radius, int normalize_xyz,
                 const float *xyz, const float *new_xyz, const float *features, const int *idx,
                 int c1, int c2, int c3, const float *w1, const float *b1, const float *w2,
                 const float *b2, const float *w3, const float *b3, float *out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* RFD_POINTNET2_H */
