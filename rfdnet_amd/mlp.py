"""The small shared MLPs of the detection stage as one kernel each (csrc/mlp_cols.hip), and feature propagation up to
its MLP as one kernel (csrc/interpolate.hip rfd_three_interpolate_cat).

`mlp_cols(x, layers)` computes, for a channel-major x (B, C0, N), the chain of 1x1 convolutions `layers` =
[(W (Cout, Cin), b (Cout), relu), ...] with eval-mode BatchNorm already folded into (W, b) (fold_bn.folded) -- exact
fp32, one fma chain per output.  Used by PointnetFPModule, VotingModule and ProposalModule at inference; training
and anything that does not fit (N % 8, widths > 1024, CPU tensors) takes the modules' own torch composition."""
import ctypes as C

import torch

from . import _lib

_wt_cache = {}


def fits(N, widths):
    """shapes rfd_mlp_cols takes (inference only: there is no backward)"""
    return N % 8 == 0 and 1 <= len(widths) - 1 <= 4 and max(widths) <= 1024 and not torch.is_grad_enabled()


def usable(x, widths):
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 3 and x.is_contiguous() and fits(x.shape[2], widths))


def _transposed(W):
    """(Cout, Cin) -> contiguous (Cin, pad4(Cout)) with zero padding (the kernel reads a lane's four consecutive
    output channels with one 16-byte load), cached per parameter version (shared by the host threads)"""
    key = (W.data_ptr(), W._version, tuple(W.shape), str(W.device))
    hit = _wt_cache.get(key)
    if hit is None:
        with _lib.BUILD_LOCK:
            hit = _wt_cache.get(key)
            if hit is None:
                cout, cin = W.shape
                wt = torch.zeros(cin, (cout + 3) // 4 * 4, dtype=W.dtype, device=W.device)
                wt[:, :cout] = W.detach().t()
                hit = (wt, W)                                 # keep the keyed tensor alive: its address is the key
                if len(_wt_cache) > 256:
                    _wt_cache.clear()
                _lib.publish(W.device)
                _wt_cache[key] = hit
    return hit[0]


def mlp_cols(x, layers):
    """x (B, C0, N) fp32 contiguous; layers = [(W (C1, C0), b (C1), relu), ...] (1 .. 4) -> (B, C_L, N)."""
    B, c0, N = x.shape
    widths = [c0] + [int(W.shape[0]) for W, _, _ in layers]
    assert usable(x, widths) and all(W.shape[1] == w for (W, _, _), w in zip(layers, widths[:-1])), widths
    n = len(layers)
    wts = [_transposed(W) for W, _, _ in layers]
    bs = [b.contiguous() for _, b, _ in layers]
    y = torch.empty(B, widths[-1], N, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = _lib.lib().rfd_mlp_cols(B, N, n, (C.c_int * (n + 1))(*widths), (C.c_void_p * n)(*[w.data_ptr() for w in wts]),
                                     (C.c_void_p * n)(*[b.data_ptr() for b in bs]),
                                     (C.c_int * n)(*[int(bool(r)) for _, _, r in layers]), x.data_ptr(), y.data_ptr(),
                                     _lib.current_stream())
    _lib.check(rc, "rfd_mlp_cols")
    return y


def interpolate_cat(known_feats, idx, dist2, skip):
    """cat([three_interpolate(known_feats (B,C,m), idx, inverse-distance weights of sqrt(dist2)), skip (B,Cs,n)], 1)
    -> (B, C + Cs, n): PointnetFPModule.forward up to its MLP (pointnet2_modules.py:383-392) in one launch."""
    B, c, m = known_feats.shape
    n = idx.shape[1]
    cs = 0 if skip is None else skip.shape[1]
    assert known_feats.is_cuda and known_feats.is_contiguous() and idx.is_contiguous() and dist2.is_contiguous()
    assert idx.dtype == torch.int32 and dist2.dtype == torch.float32 and (skip is None or skip.is_contiguous())
    out = torch.empty(B, c + cs, n, dtype=torch.float32, device=known_feats.device)
    with torch.cuda.device(known_feats.device):
        rc = _lib.lib().rfd_three_interpolate_cat(B, c, cs, m, n, known_feats.data_ptr(), idx.data_ptr(), dist2.data_ptr(),
                                                  skip.data_ptr() if skip is not None else None, out.data_ptr(),
                                                  _lib.current_stream())
    _lib.check(rc, "rfd_three_interpolate_cat")
    return out
