"""fc_pos of the skip-propagation encoder in one pass (csrc/pos_embed.hip):
out[r] = bias + mask[r] * (x[r, :d] @ W[:, :d].T + group[r // rows_per_group])."""
import torch

from . import _lib, gemm


def usable(x, W, out):
    return (x.is_cuda and x.dtype == torch.float32 and x.shape[1] <= 8 and W.shape[0] % 4 == 0
            and out.stride(1) == 1 and out.stride(0) % 4 == 0 and out.data_ptr() % 16 == 0)


def pos_embed(x, mask, W, bias, group, rows_per_group, out):
    """x (M,d) rows (row stride allowed), mask (M,), W (N, >=d) (only the first d columns are
    read), bias (N,), group (M / rows_per_group, N) contiguous, out (M,N) row view (row stride
    allowed) -> out."""
    M, d = x.shape
    N = W.shape[0]
    assert usable(x, W, out) and x.stride(1) == 1 and W.stride(1) == 1
    assert mask.is_contiguous() and mask.numel() == M and group.is_contiguous() and bias.is_contiguous()
    with torch.cuda.device(x.device):
        rc = _lib.lib().rfd_pos_embed(M, N, d, x.data_ptr(), x.stride(0), mask.data_ptr(), W.data_ptr(),
                                      W.stride(0), bias.data_ptr(), group.data_ptr(), int(rows_per_group),
                                      out.data_ptr(), out.stride(0), gemm.SA, _lib.current_stream())
    _lib.check(rc, "rfd_pos_embed")
    return out


def frag_usable(x, W, M, rows_per_group):
    return (x.is_cuda and x.dtype == torch.float32 and 1 <= x.shape[1] <= 8 and W.shape[0] % 32 == 0
            and M % 32 == 0 and rows_per_group % 32 == 0)


def pos_embed_frag(x, mask, W, bias, group, rows_per_group, out, sa):
    """the same layer written as frag rows (gemm.frag_empty layout) of relu(out) * 2^sa: `out` is a frag buffer or a
    channel window of one, (M/32, N/32, 2, 2, 64, 8) f16."""
    M, d = x.shape
    N = W.shape[0]
    assert frag_usable(x, W, M, rows_per_group) and x.stride(1) == 1 and W.stride(1) == 1
    assert mask.is_contiguous() and mask.numel() == M and group.is_contiguous() and bias.is_contiguous()
    assert out.shape[0] * 32 == M and out.shape[1] * 32 == N
    ptr, stride = gemm._frag_args(out)
    with torch.cuda.device(x.device):
        rc = _lib.lib().rfd_pos_embed_frag(M, N, d, x.data_ptr(), x.stride(0), mask.data_ptr(), W.data_ptr(),
                                           W.stride(0), bias.data_ptr(), group.data_ptr(), int(rows_per_group),
                                           ptr, stride, int(sa), _lib.current_stream())
    _lib.check(rc, "rfd_pos_embed_frag")
    return out
