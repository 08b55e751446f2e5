/* rfd_chamfer.h -- Chamfer distance (nearest neighbour both ways) for fit_mesh_to_scan
 * (models/iscnet/modules/network.py:182-303).
 *
 * C-ABI replacements of the launchers the reference's torch extension binds
 * (external/pyTorchChamferDistance/chamfer_distance/chamfer_distance.cpp:3-24):
 *   ChamferDistanceKernelLauncher      chamfer_distance.cu:139-156
 *   ChamferDistanceGradKernelLauncher  chamfer_distance.cu:188-210
 * plus a trailing stream; they return a hipError_t value instead of printing.
 * Semantics follow the reference's CPU implementation (chamfer_distance.cpp:60-180), which
 * the oracle is pinned to bit-exactly: squared distance = float x*x + y*y + z*z summed left to
 * right (no fma), lowest index wins ties.  xyz1 (b,n,3), xyz2 (b,m,3), contiguous f32. */
#ifndef RFD_CHAMFER_H
#define RFD_CHAMFER_H
#ifdef __cplusplus
extern "C" {
#endif

/* dist1 (b,n) / idx1 (b,n): nearest point of xyz2 for every point of xyz1; dist2 / idx2 (b,m)
 * the other way round.  Every element is written. */
int rfd_chamfer_forward(int b, int n, const float *xyz1, int m, const float *xyz2, float *dist1,
                        int *idx1, float *dist2, int *idx2, void *stream);

/* grad_xyz1 (b,n,3), grad_xyz2 (b,m,3) are zeroed here (cudaMemset in the reference) and then
 * accumulated with atomic adds: g = 2 grad_dist; += g (p - nn(p)) on the point, -= on its
 * neighbour.  Summation order differs from the sequential CPU loops (like the CUDA kernel). */
int rfd_chamfer_backward(int b, int n, const float *xyz1, int m, const float *xyz2,
                         const float *grad_dist1, const int *idx1, const float *grad_dist2,
                         const int *idx2, float *grad_xyz1, float *grad_xyz2, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* RFD_CHAMFER_H */
