"""`pointnet2_ops._ext` for MI355X: the nine functions the reference's pybind
module exports (external/pointnet2_ops_lib/pointnet2_ops/_ext-src/src/
bindings.cpp:6-19), same names, argument order, dtype/contiguity checks and
return conventions, backed by the HIP C-ABI library (include/rfd_pointnet2.h).

Host-side behaviour mirrored from the reference C++ wrappers:
  sampling.cpp:15-87, ball_query.cpp:8-32, group_points.cpp:12-62,
  interpolate.cpp:14-99  (CHECK_CONTIGUOUS / CHECK_IS_FLOAT / CHECK_IS_INT ->
  RuntimeError, utils.h:10-25; CPU tensors -> "CPU not supported").
Launches go to torch's current stream on the tensor's device.
"""
import torch

from .. import _lib


def _chk_contig(x, name):
    if not x.is_contiguous():
        raise RuntimeError("%s must be a contiguous tensor" % name)


def _chk_float(x, name):
    if x.dtype != torch.float32:
        raise RuntimeError("%s must be a float tensor" % name)


def _chk_int(x, name):
    if x.dtype != torch.int32:
        raise RuntimeError("%s must be an int tensor" % name)


def _chk_cuda(x, name):
    if not x.is_cuda:
        raise RuntimeError("%s must be a CUDA tensor" % name)


def _need_gpu(x):
    if not x.is_cuda:
        raise RuntimeError("CPU not supported")       # sampling.cpp:34 et al.


def _call(fn_name, x, *args):
    with torch.cuda.device(x.device):
        rc = getattr(_lib.lib(), fn_name)(*args, _lib.current_stream())
    _lib.check(rc, fn_name)


def gather_points(points, idx):
    """(B,C,N) f32, (B,M) i32 -> (B,C,M).  sampling.cpp:15-38."""
    _chk_contig(points, "points"); _chk_contig(idx, "idx")
    _chk_float(points, "points"); _chk_int(idx, "idx")
    if points.is_cuda:
        _chk_cuda(idx, "idx")
    _need_gpu(points)
    B, Cc, N = points.shape
    M = idx.shape[1]
    out = torch.empty((B, Cc, M), dtype=torch.float32, device=points.device)
    _call("gather_points_kernel_wrapper", points, B, Cc, N, M,
          points.data_ptr(), idx.data_ptr(), out.data_ptr())
    return out


def gather_points_grad(grad_out, idx, n):
    """sampling.cpp:40-64."""
    _chk_contig(grad_out, "grad_out"); _chk_contig(idx, "idx")
    _chk_float(grad_out, "grad_out"); _chk_int(idx, "idx")
    if grad_out.is_cuda:
        _chk_cuda(idx, "idx")
    _need_gpu(grad_out)
    B, Cc, M = grad_out.shape
    out = torch.zeros((B, Cc, n), dtype=torch.float32, device=grad_out.device)
    _call("gather_points_grad_kernel_wrapper", grad_out, B, Cc, n, M,
          grad_out.data_ptr(), idx.data_ptr(), out.data_ptr())
    return out


def furthest_point_sampling(points, nsamples):
    """(B,N,3) f32 -> (B,nsamples) i32.  sampling.cpp:66-87."""
    _chk_contig(points, "points"); _chk_float(points, "points")
    _need_gpu(points)
    B, N = points.shape[0], points.shape[1]
    out = torch.zeros((B, nsamples), dtype=torch.int32, device=points.device)
    tmp = torch.empty((B, N), dtype=torch.float32, device=points.device)
    _call("furthest_point_sampling_kernel_wrapper", points, B, N, nsamples,
          points.data_ptr(), tmp.data_ptr(), out.data_ptr())
    return out


def three_nn(unknowns, knows):
    """(B,n,3), (B,m,3) -> [dist2 (B,n,3) f32, idx (B,n,3) i32].  interpolate.cpp:14-40."""
    _chk_contig(unknowns, "unknowns"); _chk_contig(knows, "knows")
    _chk_float(unknowns, "unknowns"); _chk_float(knows, "knows")
    if unknowns.is_cuda:
        _chk_cuda(knows, "knows")
    _need_gpu(unknowns)
    B, n = unknowns.shape[0], unknowns.shape[1]
    m = knows.shape[1]
    idx = torch.empty((B, n, 3), dtype=torch.int32, device=unknowns.device)
    dist2 = torch.empty((B, n, 3), dtype=torch.float32, device=unknowns.device)
    _call("three_nn_kernel_wrapper", unknowns, B, n, m, unknowns.data_ptr(),
          knows.data_ptr(), dist2.data_ptr(), idx.data_ptr())
    return [dist2, idx]


def three_interpolate(points, idx, weight):
    """(B,C,m), (B,n,3) i32, (B,n,3) f32 -> (B,C,n).  interpolate.cpp:42-70."""
    _chk_contig(points, "points"); _chk_contig(idx, "idx"); _chk_contig(weight, "weight")
    _chk_float(points, "points"); _chk_int(idx, "idx"); _chk_float(weight, "weight")
    if points.is_cuda:
        _chk_cuda(idx, "idx"); _chk_cuda(weight, "weight")
    _need_gpu(points)
    B, Cc, m = points.shape
    n = idx.shape[1]
    out = torch.empty((B, Cc, n), dtype=torch.float32, device=points.device)
    _call("three_interpolate_kernel_wrapper", points, B, Cc, m, n,
          points.data_ptr(), idx.data_ptr(), weight.data_ptr(), out.data_ptr())
    return out


def three_interpolate_grad(grad_out, idx, weight, m):
    """interpolate.cpp:72-99."""
    _chk_contig(grad_out, "grad_out"); _chk_contig(idx, "idx"); _chk_contig(weight, "weight")
    _chk_float(grad_out, "grad_out"); _chk_int(idx, "idx"); _chk_float(weight, "weight")
    if grad_out.is_cuda:
        _chk_cuda(idx, "idx"); _chk_cuda(weight, "weight")
    _need_gpu(grad_out)
    B, Cc, n = grad_out.shape
    out = torch.zeros((B, Cc, m), dtype=torch.float32, device=grad_out.device)
    _call("three_interpolate_grad_kernel_wrapper", grad_out, B, Cc, n, m,
          grad_out.data_ptr(), idx.data_ptr(), weight.data_ptr(), out.data_ptr())
    return out


def ball_query(new_xyz, xyz, radius, nsample):
    """(B,M,3), (B,N,3) -> (B,M,nsample) i32.  ball_query.cpp:8-32."""
    _chk_contig(new_xyz, "new_xyz"); _chk_contig(xyz, "xyz")
    _chk_float(new_xyz, "new_xyz"); _chk_float(xyz, "xyz")
    if new_xyz.is_cuda:
        _chk_cuda(xyz, "xyz")
    _need_gpu(new_xyz)
    B, N = xyz.shape[0], xyz.shape[1]
    M = new_xyz.shape[1]
    # the kernel writes every element (zeros for empty balls): no zero-fill pass
    idx = torch.empty((B, M, nsample), dtype=torch.int32, device=new_xyz.device)
    _call("query_ball_point_kernel_wrapper", new_xyz, B, N, M, float(radius),
          int(nsample), new_xyz.data_ptr(), xyz.data_ptr(), idx.data_ptr())
    return idx


def group_points(points, idx):
    """(B,C,N), (B,M,ns) i32 -> (B,C,M,ns).  group_points.cpp:12-36."""
    _chk_contig(points, "points"); _chk_contig(idx, "idx")
    _chk_float(points, "points"); _chk_int(idx, "idx")
    if points.is_cuda:
        _chk_cuda(idx, "idx")
    _need_gpu(points)
    B, Cc, N = points.shape
    M, ns = idx.shape[1], idx.shape[2]
    out = torch.empty((B, Cc, M, ns), dtype=torch.float32, device=points.device)
    _call("group_points_kernel_wrapper", points, B, Cc, N, M, ns,
          points.data_ptr(), idx.data_ptr(), out.data_ptr())
    return out


def group_points_grad(grad_out, idx, n):
    """group_points.cpp:38-62."""
    _chk_contig(grad_out, "grad_out"); _chk_contig(idx, "idx")
    _chk_float(grad_out, "grad_out"); _chk_int(idx, "idx")
    if grad_out.is_cuda:
        _chk_cuda(idx, "idx")
    _need_gpu(grad_out)
    B, Cc, M, ns = grad_out.shape
    out = torch.zeros((B, Cc, n), dtype=torch.float32, device=grad_out.device)
    _call("group_points_grad_kernel_wrapper", grad_out, B, Cc, n, M, ns,
          grad_out.data_ptr(), idx.data_ptr(), out.data_ptr())
    return out


# ---- fused forms (no reference counterpart; see include/rfd_pointnet2.h) ------

def furthest_point_sampling_gather(points, nsamples):
    """FPS that also returns the sampled centres (B,nsamples,3)."""
    _chk_contig(points, "points"); _chk_float(points, "points")
    _need_gpu(points)
    B, N = points.shape[0], points.shape[1]
    out = torch.zeros((B, nsamples), dtype=torch.int32, device=points.device)
    new_xyz = torch.empty((B, nsamples, 3), dtype=torch.float32, device=points.device)
    tmp = torch.empty((B, N), dtype=torch.float32, device=points.device)
    _call("rfd_furthest_point_sampling_gather", points, B, N, nsamples,
          points.data_ptr(), tmp.data_ptr(), out.data_ptr(), new_xyz.data_ptr())
    return out, new_xyz


def group_concat(xyz, new_xyz, features, idx, radius, normalize_xyz, use_xyz,
                 ret_grouped_xyz):
    """QueryAndGroup epilogue in one pass (pointnet2_utils.py:333-344)."""
    _chk_contig(xyz, "xyz"); _chk_contig(new_xyz, "new_xyz"); _chk_contig(idx, "idx")
    _chk_float(xyz, "xyz"); _chk_float(new_xyz, "new_xyz"); _chk_int(idx, "idx")
    _need_gpu(xyz)
    B, N = xyz.shape[0], xyz.shape[1]
    M, ns = idx.shape[1], idx.shape[2]
    Cc = 0
    fptr = None
    if features is not None:
        _chk_contig(features, "features"); _chk_float(features, "features")
        Cc = features.shape[1]
        fptr = features.data_ptr()
    ctot = (3 if use_xyz else 0) + Cc
    out = torch.empty((B, ctot, M, ns), dtype=torch.float32, device=xyz.device)
    gx = (torch.empty((B, 3, M, ns), dtype=torch.float32, device=xyz.device)
          if ret_grouped_xyz else None)
    _call("rfd_group_concat", xyz, B, Cc, N, M, ns, float(radius),
          int(bool(normalize_xyz)), int(bool(use_xyz)), xyz.data_ptr(),
          new_xyz.data_ptr(), fptr, idx.data_ptr(), out.data_ptr(),
          gx.data_ptr() if gx is not None else None)
    return out, gx
