"""One ResnetBlockFC of the skip-propagation point encoder as a single fused
kernel (csrc/resblock.hip; reference: models/iscnet/modules/layers.py:39-48 inside
ResnetPointnet.forward, layers.py:364-392).

    out = Ws relu(x) + W1 relu(W0 relu(x) + g0[group]) + gs[group]

x (M, k_in) fp32 rows, k_in = 256 or 512; W0 / Ws = the first k_in columns of
fc_0.weight / shortcut.weight; g0, gs (groups, 256) carry the biases and the
pooled-context share of the block input (see ResnetPointnet.forward_factored).
Weights are split into f16 (hi, lo), re-laid in consumption order and cached per
parameter version."""
import torch

from . import _lib, occ_fold

HIDDEN = 256
TILE = 128


def usable(x, rows_per_group):
    M, k_in = x.shape
    return (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and k_in in (256, 512)
            and M % TILE == 0 and rows_per_group % TILE == 0 and x.data_ptr() % 16 == 0)


def _packed(block, k_in):
    """Packed weight stream of `block`, cached ON the block (a cache keyed by id() or
    data_ptr() alone would alias a later module that reuses the address)."""
    ps = (block.fc_0.weight, block.shortcut.weight, block.fc_1.weight)
    key = (k_in,) + tuple((p.data_ptr(), p._version, str(p.device)) for p in ps)
    hit = block.__dict__.get('_rfd_resblock')
    if hit is None or hit[0] != key:
        w0, ws, w1 = (p.detach() for p in ps)
        assert w0.shape[0] == HIDDEN and ws.shape[0] == HIDDEN and tuple(w1.shape) == (HIDDEN, HIDDEN)
        assert w0.shape[1] == ws.shape[1] and w0.shape[1] >= k_in and w0.is_contiguous() and ws.is_contiguous()
        kw0 = occ_fold.choose_kw([w0[:, :k_in]])
        kw1 = occ_fold.choose_kw([ws[:, :k_in], w1])       # Ws and W1 accumulate into the same registers
        buf = torch.empty(_lib.lib().rfd_resblock_packed_bytes(k_in), dtype=torch.uint8, device=w0.device)
        with torch.cuda.device(w0.device):
            rc = _lib.lib().rfd_resblock_pack(k_in, w0.shape[1], w0.data_ptr(), ws.data_ptr(),
                                              w1.contiguous().data_ptr(), kw0, kw1, buf.data_ptr(),
                                              _lib.current_stream())
        _lib.check(rc, "rfd_resblock_pack")
        hit = (key, buf, kw0, kw1)
        block.__dict__['_rfd_resblock'] = hit
    return hit[1:]


def forward(block, x, g0, gs, rows_per_group, out=None):
    """block: ResnetBlockFC (size_h = size_out = 256, with shortcut); x (M, k_in);
    g0, gs (M / rows_per_group, 256) fp32 contiguous -> (M, 256)."""
    M, k_in = x.shape
    assert usable(x, rows_per_group)
    packed, kw0, kw1 = _packed(block, k_in)
    G = M // rows_per_group
    assert M % rows_per_group == 0 and tuple(g0.shape) == (G, HIDDEN) and tuple(gs.shape) == (G, HIDDEN)
    assert g0.is_contiguous() and gs.is_contiguous() and g0.dtype == torch.float32 and gs.dtype == torch.float32
    if out is None:
        out = torch.empty(M, HIDDEN, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = _lib.lib().rfd_resblock_f16x3(M, k_in, int(rows_per_group), x.data_ptr(), packed.data_ptr(),
                                           g0.data_ptr(), gs.data_ptr(), out.data_ptr(), kw0, kw1,
                                           _lib.current_stream())
    _lib.check(rc, "rfd_resblock_f16x3")
    return out
