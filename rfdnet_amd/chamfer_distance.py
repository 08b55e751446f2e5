"""Chamfer distance with the reference's names
(external/pyTorchChamferDistance/chamfer_distance/chamfer_distance.py:9-66):
`ChamferDistanceFunction.apply(xyz1, xyz2) -> (dist1, dist2)` (squared distances to the
nearest point of the other set), differentiable w.r.t. both point sets, and the
`ChamferDistance` module.  HIP kernels: csrc/chamfer.hip.  CPU tensors raise (the reference
falls back to its C++ loops; this package has no CPU path)."""
import torch

from . import _lib


class ChamferDistanceFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz1, xyz2):
        if not (xyz1.is_cuda and xyz2.is_cuda):
            raise RuntimeError("CPU not supported")
        xyz1 = xyz1.contiguous().float()
        xyz2 = xyz2.contiguous().float()
        b, n, _ = xyz1.size()
        m = xyz2.size(1)
        dist1 = torch.empty(b, n, device=xyz1.device)
        dist2 = torch.empty(b, m, device=xyz1.device)
        idx1 = torch.empty(b, n, dtype=torch.int32, device=xyz1.device)
        idx2 = torch.empty(b, m, dtype=torch.int32, device=xyz1.device)
        with torch.cuda.device(xyz1.device):
            rc = _lib.lib().rfd_chamfer_forward(b, n, xyz1.data_ptr(), m, xyz2.data_ptr(), dist1.data_ptr(),
                                                idx1.data_ptr(), dist2.data_ptr(), idx2.data_ptr(),
                                                _lib.current_stream())
        _lib.check(rc, "rfd_chamfer_forward")
        ctx.save_for_backward(xyz1, xyz2, idx1, idx2)
        ctx.mark_non_differentiable(idx1, idx2)
        return dist1, dist2

    @staticmethod
    def backward(ctx, graddist1, graddist2):
        xyz1, xyz2, idx1, idx2 = ctx.saved_tensors
        graddist1 = graddist1.contiguous()
        graddist2 = graddist2.contiguous()
        b, n, _ = xyz1.size()
        m = xyz2.size(1)
        g1 = torch.empty_like(xyz1)
        g2 = torch.empty_like(xyz2)
        with torch.cuda.device(xyz1.device):
            rc = _lib.lib().rfd_chamfer_backward(b, n, xyz1.data_ptr(), m, xyz2.data_ptr(), graddist1.data_ptr(),
                                                 idx1.data_ptr(), graddist2.data_ptr(), idx2.data_ptr(),
                                                 g1.data_ptr(), g2.data_ptr(), _lib.current_stream())
        _lib.check(rc, "rfd_chamfer_backward")
        return g1, g2


class ChamferDistance(torch.nn.Module):
    def forward(self, xyz1, xyz2):
        return ChamferDistanceFunction.apply(xyz1, xyz2)


def nearest(xyz1, xyz2):
    """(dist1, idx1, dist2, idx2) without autograd (tests, diagnostics)."""
    xyz1 = xyz1.contiguous().float()
    xyz2 = xyz2.contiguous().float()
    b, n, _ = xyz1.size()
    m = xyz2.size(1)
    dist1 = torch.empty(b, n, device=xyz1.device)
    dist2 = torch.empty(b, m, device=xyz1.device)
    idx1 = torch.empty(b, n, dtype=torch.int32, device=xyz1.device)
    idx2 = torch.empty(b, m, dtype=torch.int32, device=xyz1.device)
    with torch.cuda.device(xyz1.device):
        rc = _lib.lib().rfd_chamfer_forward(b, n, xyz1.data_ptr(), m, xyz2.data_ptr(), dist1.data_ptr(),
                                            idx1.data_ptr(), dist2.data_ptr(), idx2.data_ptr(),
                                            _lib.current_stream())
    _lib.check(rc, "rfd_chamfer_forward")
    return dist1, idx1, dist2, idx2
