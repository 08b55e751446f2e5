"""Inference-time folding of BatchNorm (eval mode) into the preceding 1x1
convolution / linear layer, and a row-major linear helper that routes layers
that tile onto the split-precision GEMM kernel (csrc/gemm_f16x3.hip) with the
bias + ReLU fused in its epilogue.

  BN(Wx + b) = (s*W) x + (s*(b - mean) + beta),   s = gamma / sqrt(var + eps)
"""
import torch
import torch.nn.functional as F

from . import gemm


def folded(layer, bn=None):
    """(W (N,K), b (N)) of `layer` (Conv1d k=1 or Linear) with `bn` folded in.
    Cached on the layer, keyed by the parameter versions."""
    w = layer.weight
    key = (w._version, w.data_ptr(), None if layer.bias is None else layer.bias._version,
           None if bn is None else (bn.running_mean._version, bn.running_var._version,
                                    None if bn.weight is None else bn.weight._version,
                                    None if bn.bias is None else bn.bias._version))
    hit = layer.__dict__.get('_folded')
    if hit is not None and hit[0] == key:
        return hit[1], hit[2]
    from . import _lib
    with _lib.BUILD_LOCK:                    # shared across host threads: built once, published before it is stored
        hit = layer.__dict__.get('_folded')
        if hit is not None and hit[0] == key:
            return hit[1], hit[2]
        return _fold_now(layer, bn, w, key)


def _fold_now(layer, bn, w, key):
    from . import _lib
    W = w.detach().reshape(w.shape[0], -1)
    b = layer.bias.detach() if layer.bias is not None else torch.zeros(W.shape[0], device=W.device, dtype=W.dtype)
    if bn is not None:
        s = torch.rsqrt(bn.running_var + bn.eps)
        if bn.weight is not None:
            s = s * bn.weight.detach()
        beta = bn.bias.detach() if bn.bias is not None else torch.zeros_like(s)
        W = W * s[:, None]
        b = (b - bn.running_mean) * s + beta
    W, b = W.contiguous(), b.contiguous()
    if W.is_cuda:
        _lib.publish(W.device)
    layer.__dict__['_folded'] = (key, W, b)
    return W, b


def linear_rows_pooled(x, W, b, rows_per_group, relu=True):
    """max over each group of rows_per_group rows of relu?(x W^T + b) -> (M / rows_per_group, N),
    without writing the (M, N) product when the fused epilogue is available."""
    M, K = x.shape
    N = W.shape[0]
    if gemm.usable(M, N, K, x) and gemm.pool_usable(M, N, K, rows_per_group):
        pool = torch.full((M // rows_per_group, N), 0.0 if relu else float('-inf'), device=x.device, dtype=x.dtype)
        gemm.linear(x, W, bias=b, relu_out=relu, rows_per_group=rows_per_group, pool=pool, store=False,
                    pool_signed=not relu)
        return pool
    return linear_rows(x, W, b, relu=relu).view(-1, rows_per_group, N).max(dim=1)[0]


def linear_rows(x, W, b, relu=False, gbias=None, rows_per_group=1):
    """x (M,K) fp32 rows -> relu?(x W^T + b [+ gbias[row // rows_per_group]])."""
    M, K = x.shape
    N = W.shape[0]
    if gemm.usable(M, N, K, x) and (gbias is None or gbias.is_contiguous()):
        return gemm.linear(x, W, bias=b, gbias=gbias, rows_per_group=rows_per_group, relu_out=relu)
    y = F.linear(x, W, b)
    if gbias is not None:
        y = (y.view(-1, rows_per_group, N) + gbias.unsqueeze(1)).view(M, N)
    return y.relu_() if relu else y
