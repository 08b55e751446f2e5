"""The PointNet feature chains of the skip-propagation nets as one kernel each (csrc/pointseg_chain.hip):
[d -> 64, ReLU] -> 64 -> 128, ReLU -> 128 -> C3 [, ReLU] -> max over the P points of a proposal (C3 = 1024 for
PointSeg's encoders, 256 for STN_Group's STN3d; any multiple of 64 up to 1024), on the split-f16
matrix-core arithmetic of gemm.py (fp32-class).  Weights are (W (N,K), b (N)) with BatchNorm already folded in
(fold_bn.folded); they are split / re-laid once and cached per parameter version."""
import os

import torch

from . import _lib, gemm, occ_fold

_cache = {}


def usable(x, P, d_in):
    """x (M, d) fp32 CUDA rows, P points per proposal."""
    M = x.shape[0]
    if os.environ.get("RFD_NO_CHAIN") == "1":           # A/B timing against the GEMM-per-layer path
        return False
    ok = (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1 and P % 512 == 0 and M % P == 0)
    if d_in <= 8:
        return ok
    return ok and d_in == 64 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0


def _cached(key, build):
    """shared by every host thread running the same network: a miss is built under the library's build lock and published
    (building stream drained) before it is stored (_lib.BUILD_LOCK)"""
    hit = _cache.get(key)
    if hit is None:
        with _lib.BUILD_LOCK:
            hit = _cache.get(key)
            if hit is None:
                hit = build()
                if len(_cache) > 64:
                    _cache.clear()
                _lib.publish(hit[0].device)
                _cache[key] = hit
    return hit


def _packed(mode, layers):
    (W1, _), (W2, _), (W3, _) = layers
    key = tuple((None if w is None else (w.data_ptr(), w._version, tuple(w.shape))) for w in (W1, W2, W3)) + (mode,)

    def build():
        c3 = W3.shape[0]
        assert tuple(W2.shape) == (128, 64) and W3.shape[1] == 128 and c3 % 64 == 0 and 64 <= c3 <= 1024, \
            (W2.shape, W3.shape)
        sw1 = occ_fold.choose_kw([W1]) if mode == 2 else 0
        sw2, sw3 = occ_fold.choose_kw([W2]), occ_fold.choose_kw([W3])
        buf = torch.empty(_lib.lib().rfd_chain_packed_bytes_n(c3), dtype=torch.uint8, device=W2.device)
        w1c = W1.contiguous() if mode == 2 else None
        w2c, w3c = W2.contiguous(), W3.contiguous()
        with torch.cuda.device(W2.device):
            rc = _lib.lib().rfd_chain_pack_n(mode, c3, w1c.data_ptr() if w1c is not None else None, w2c.data_ptr(),
                                             w3c.data_ptr(), sw1, sw2, sw3, buf.data_ptr(), _lib.current_stream())
        _lib.check(rc, "rfd_chain_pack")
        torch.cuda.current_stream(W2.device).synchronize()          # w?c may be temporaries
        return (buf, sw1, sw2, sw3, (W1, W2, W3))                    # keep the keyed tensors alive
    return _cached(key, build)


def chain_pool(x, layer1, layer2, layer3, P, relu3):
    """x (M, d); layer1 = (W1 (64,d), b1) or None (x is already the 64-wide feature); layer2 = (W2 (128,64), b2);
    layer3 = (W3 (C3,128), b3), C3 a multiple of 64 up to 1024 -> (M / P, C3) = max over each proposal's points of the
    chain's output."""
    M, d = x.shape
    if layer1 is None:
        mode, layer1 = 0, (None, None)
    else:
        mode = 1 if d <= 8 else 2
        assert tuple(layer1[0].shape) == (64, d), layer1[0].shape
    assert usable(x, P, d)
    buf, sw1, sw2, sw3, _ = _packed(mode, (layer1, layer2, layer3))
    c3 = layer3[0].shape[0]
    out = torch.empty(M // P, c3, dtype=torch.float32, device=x.device)
    w1raw = layer1[0].contiguous() if mode == 1 else None
    b1 = layer1[1].contiguous() if mode else None
    b2, b3 = layer2[1].contiguous(), layer3[1].contiguous()
    with torch.cuda.device(x.device):
        rc = _lib.lib().rfd_chain_pool_n(mode, c3, M, P, d, x.data_ptr(), x.stride(0), buf.data_ptr(),
                                       w1raw.data_ptr() if w1raw is not None else None,
                                       b1.data_ptr() if b1 is not None else None, b2.data_ptr(), b3.data_ptr(),
                                       int(bool(relu3)), gemm.SA, sw1, sw2, sw3, out.data_ptr(), _lib.current_stream())
    _lib.check(rc, "rfd_chain_pool")
    return out


def head_usable(x, P, n_cls):
    return (os.environ.get("RFD_NO_CHAIN") != "1" and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2
            and x.shape[1] == 64 and x.stride(1) == 1 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0
            and P % 128 == 0 and x.shape[0] % P == 0 and 1 <= n_cls <= 2)


def _head_packed(Wa, Wb, Wc):
    key = tuple((w.data_ptr(), w._version, tuple(w.shape)) for w in (Wa, Wb, Wc)) + ("head",)

    def build():
        assert tuple(Wa.shape) == (512, 64) and tuple(Wb.shape) == (256, 512) and tuple(Wc.shape) == (128, 256)
        swa, swb, swc = (occ_fold.choose_kw([w]) for w in (Wa, Wb, Wc))
        buf = torch.empty(_lib.lib().rfd_head_packed_bytes(), dtype=torch.uint8, device=Wa.device)
        wa, wb, wc = Wa.contiguous(), Wb.contiguous(), Wc.contiguous()
        with torch.cuda.device(Wa.device):
            rc = _lib.lib().rfd_head_pack(wa.data_ptr(), wb.data_ptr(), wc.data_ptr(), swa, swb, swc, buf.data_ptr(),
                                          _lib.current_stream())
        _lib.check(rc, "rfd_head_pack")
        torch.cuda.current_stream(Wa.device).synchronize()
        return (buf, swa, swb, swc, (Wa, Wb, Wc))
    return _cached(key, build)


def head_scores(x, P, Wa, gbias, layer_b, layer_c, Wd, bd):
    """PointSeg's per-point head in one kernel: x (M,64) point features; Wa (512,64) = conv1's point-feature columns,
    gbias (M/P,512) = conv1's bias + its global-feature share per proposal; layer_b = (Wb (256,512), bb), layer_c =
    (Wc (128,256), bc) with BatchNorm folded; Wd (n_cls,128), bd (n_cls,) -> raw class scores (M, n_cls)."""
    M = x.shape[0]
    n_cls = Wd.shape[0]
    assert head_usable(x, P, n_cls) and tuple(gbias.shape) == (M // P, 512)
    buf, swa, swb, swc, _ = _head_packed(Wa, layer_b[0], layer_c[0])
    out = torch.empty(M, n_cls, dtype=torch.float32, device=x.device)
    gb, bb, bc, wd, b_d = (t.contiguous() for t in (gbias, layer_b[1], layer_c[1], Wd, bd))
    with torch.cuda.device(x.device):
        rc = _lib.lib().rfd_head_scores(M, P, x.data_ptr(), x.stride(0), buf.data_ptr(), gb.data_ptr(), bb.data_ptr(),
                                        bc.data_ptr(), wd.data_ptr(), b_d.data_ptr(), n_cls, gemm.SA, swa, swb, swc,
                                        out.data_ptr(), _lib.current_stream())
    _lib.check(rc, "rfd_head_scores")
    return out
