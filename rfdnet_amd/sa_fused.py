"""Fused set-abstraction MLP (csrc/sa_fused.hip): grouping + 3 x [1x1 conv + BN(eval) +
ReLU] + max over the neighbours in one kernel (pointnet2_modules.py:219-255 after the
ball query).  The BatchNorms are folded into the convolutions and the weights re-laid in
MFMA-fragment order ONCE per parameter version (cached on the nn.Sequential)."""
import os

import torch

from . import _lib

SHAPES = {(4, 64, 64, 128), (131, 128, 128, 256), (259, 128, 128, 256), (259, 128, 128, 128)}


def _fold(conv, bn):
    w = conv.weight.detach().reshape(conv.weight.shape[0], -1)
    b = conv.bias.detach() if conv.bias is not None else torch.zeros(w.shape[0], device=w.device, dtype=w.dtype)
    s = torch.rsqrt(bn.running_var + bn.eps)
    if bn.weight is not None:
        s = s * bn.weight.detach()
    beta = bn.bias.detach() if bn.bias is not None else torch.zeros_like(s)
    return w * s[:, None], ((b - bn.running_mean) * s + beta).contiguous()


def _korder_first(kj):
    j = torch.arange(kj).view(kj, 1)
    kh = torch.arange(2).view(1, 2)
    return 2 * j + kh                                              # (kj, 2)


def _korder_next(kj):
    """k-step j of a later layer reads the previous layer's accumulator register j & 15 of
    block j >> 4: channels 32 (j >> 4) + 8 ((j & 15) >> 2) + 4 kh + (j & 3)."""
    j = torch.arange(kj).view(kj, 1)
    kh = torch.arange(2).view(1, 2)
    return 32 * (j >> 4) + 8 * ((j & 15) >> 2) + 4 * kh + (j & 3)


def pack_layer(w, korder):
    """w (C, K) -> [C/32][KJ/4][64 lanes][4]: element (b, j4, lane, e) = w[32b + (lane & 31),
    korder[4 j4 + e, lane >> 5]] (zero where korder points past K)."""
    C, K = w.shape
    kj = korder.shape[0]
    assert C % 32 == 0 and kj % 4 == 0
    wz = torch.cat([w, torch.zeros(C, 1, device=w.device, dtype=w.dtype)], dim=1)      # column K = 0
    ko = korder.to(w.device).clamp(max=K)                                               # (kj, 2)
    lane = torch.arange(64, device=w.device)
    rows = (32 * torch.arange(C // 32, device=w.device).view(-1, 1, 1, 1) + (lane & 31).view(1, 1, 64, 1))
    cols = ko.view(kj // 4, 4, 2)[:, :, lane >> 5].permute(0, 2, 1).unsqueeze(0)       # (1, kj/4, 64, 4)
    return wz[rows.expand(-1, kj // 4, -1, 4), cols.expand(C // 32, -1, -1, -1)].contiguous()


def _packed(mlp):
    convs = [m for m in mlp if isinstance(m, torch.nn.Conv2d)]
    bns = [m for m in mlp if isinstance(m, torch.nn.BatchNorm2d)]
    key = tuple((p.data_ptr(), p._version) for m in convs + bns for p in list(m.parameters()) + list(m.buffers()))
    hit = mlp.__dict__.get('_rfd_sa_packed')
    if hit is None or hit[0] != key:
        from . import _lib
        with _lib.BUILD_LOCK:          # shared across host threads: built once, published before it is stored
            hit = mlp.__dict__.get('_rfd_sa_packed')
            if hit is None or hit[0] != key:
                out = []
                for li, (cv, bn) in enumerate(zip(convs, bns)):
                    w, b = _fold(cv, bn)
                    K = w.shape[1]
                    korder = _korder_first(((K + 7) // 8) * 4) if li == 0 else _korder_next(K // 2)
                    out += [pack_layer(w, korder), b]
                hit = (key, out)
                if out[0].is_cuda:
                    _lib.publish(out[0].device)
                mlp.__dict__['_rfd_sa_packed'] = hit
    return hit[1]


def usable(mlp, features, nsample, pooling, use_xyz):
    if os.environ.get("RFD_SA_UNFUSED"):            # A/B switch: take the op composition instead
        return False
    if torch.is_grad_enabled() or mlp.training or pooling != 'max' or not use_xyz or features is None:
        return False
    convs = [m for m in mlp if isinstance(m, torch.nn.Conv2d)]
    bns = [m for m in mlp if isinstance(m, torch.nn.BatchNorm2d)]
    if len(convs) != 3 or len(bns) != 3 or nsample not in (16, 32, 64) or not features.is_cuda:
        return False
    return (convs[0].in_channels, convs[0].out_channels, convs[1].out_channels, convs[2].out_channels) in SHAPES \
        and convs[0].in_channels == 3 + features.shape[1]


def forward(mlp, xyz, new_xyz, features, idx, radius, normalize_xyz):
    """xyz (B,N,3), new_xyz (B,M,3), features (B,C,N), idx (B,M,ns) i32 -> (B, C3, M)."""
    w1, b1, w2, b2, w3, b3 = _packed(mlp)
    B, N, _ = xyz.shape
    M, ns = idx.shape[1], idx.shape[2]
    C = features.shape[1]
    c1, c2, c3 = b1.numel(), b2.numel(), b3.numel()
    for t in (xyz, new_xyz, features, idx):
        assert t.is_contiguous()
    out = torch.empty(B, c3, M, dtype=torch.float32, device=xyz.device)
    with torch.cuda.device(xyz.device):
        rc = _lib.lib().rfd_sa_fused(B, N, M, ns, C, float(radius), int(bool(normalize_xyz)), xyz.data_ptr(),
                                     new_xyz.data_ptr(), features.data_ptr(), idx.data_ptr(), c1, c2, c3,
                                     w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(),
                                     w3.data_ptr(), b3.data_ptr(), out.data_ptr(), _lib.current_stream())
    _lib.check(rc, "rfd_sa_fused")
    return out
