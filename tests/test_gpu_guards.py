"""GPU (-m gpu): guarded-buffer runs of the kernels (memory-error detection; the reference has
none).  Every device tensor the host side allocates for a kernel's OUTPUT (torch.empty / zeros / full inside the
wrappers) is carved out of a larger allocation whose 4 KiB on either side carries a poison pattern; after the hot
path has run -- point ops, fused SA layer, skip propagation, MISE rounds with the scatter-fused decoder, marching
cubes -- every guard must still hold the pattern.  An out-of-bounds store of a few elements (the classic ragged-tile
mistake) lands in a guard instead of in somebody else's tensor."""
import contextlib

import numpy as np
import pytest
import torch

from rfdnet_amd import synthetic

pytestmark = pytest.mark.gpu
GUARD = 4096
POISON = 0xA5


class Guards(object):
    def __init__(self):
        self.live = []
        self._orig = {name: getattr(torch, name) for name in ("empty", "zeros", "full")}

    def _alloc(self, shape, dtype, device, fill=None):
        n = int(np.prod(shape)) if len(shape) else 1
        nbytes = n * self._orig["empty"]((), dtype=dtype).element_size()
        pad = (-nbytes) % 256
        base = self._orig["full"]((GUARD + nbytes + pad + GUARD,), POISON, dtype=torch.uint8, device=device)
        view = base[GUARD:GUARD + nbytes].view(dtype).view(shape)
        if fill is not None:
            view.fill_(fill)
        self.live.append((base, nbytes, tuple(shape), dtype))
        return view

    def _wrap(self, name):
        orig = self._orig[name]

        def fn(*args, **kw):
            dev = kw.get("device")
            if dev is None or not str(dev).startswith("cuda") or kw.get("pin_memory") or kw.get("out") is not None:
                return orig(*args, **kw)
            dtype = kw.get("dtype") or torch.float32
            if name == "full":
                shape, fill = args[0], args[1]
            else:
                shape, fill = (args[0] if len(args) == 1 and isinstance(args[0], (tuple, list, torch.Size)) else args), \
                    (0 if name == "zeros" else None)
            shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list, torch.Size)) else (shape,)))
            if any(s < 0 for s in shape) or dtype in (torch.bool,):
                return orig(*args, **kw)
            return self._alloc(shape, dtype, torch.device(dev), fill)
        return fn

    @contextlib.contextmanager
    def active(self):
        for name in ("empty", "zeros", "full"):
            setattr(torch, name, self._wrap(name))
        try:
            yield self
        finally:
            for name, orig in self._orig.items():
                setattr(torch, name, orig)

    def check(self):
        torch.cuda.synchronize()
        bad = []
        for base, nbytes, shape, dtype in self.live:
            lo = base[:GUARD]
            hi = base[GUARD + nbytes:]
            if not bool((lo == POISON).all()) or not bool((hi == POISON).all()):
                first_hi = int(torch.nonzero(hi != POISON)[0]) if not bool((hi == POISON).all()) else None
                bad.append((shape, dtype, "below" if not bool((lo == POISON).all()) else "above", first_hi))
        return len(self.live), bad


def test_point_ops_stay_inside_their_outputs(hip):
    from rfdnet_amd.pointnet2_ops import _ext
    g = Guards()
    pc = synthetic.synthetic_scene(seed=4, n_raw=30000, n_points=20011)          # ragged sizes on purpose
    xyz = torch.from_numpy(np.ascontiguousarray(pc[None, :, :3])).cuda()
    feats = torch.from_numpy(np.ascontiguousarray(pc[None, :, 3:].transpose(0, 2, 1))).cuda()
    with g.active():
        inds = _ext.furthest_point_sampling(xyz, 517)
        ctr = _ext.gather_points(xyz.transpose(1, 2).contiguous(), inds).transpose(1, 2).contiguous()
        idx = _ext.ball_query(ctr, xyz, 0.25, 37)
        grouped = _ext.group_points(feats, idx)
        d2, i3 = _ext.three_nn(xyz, ctr)
        w = torch.ones_like(d2) / 3
        _ext.three_interpolate(grouped[:, :, :, 0].contiguous(), i3, w)
    n, bad = g.check()
    hip.device_status()
    assert n >= 5, n
    assert bad == [], bad


@pytest.mark.parametrize("steps", [0, 1, 2])
def test_scene_pipeline_stays_inside_its_outputs(hip, steps):
    """backbone (fused SA layers) -> voting -> proposals -> skip propagation (split GEMMs, pos_embed, pooled
    epilogues) -> MISE (count / collect / scatter-fused decode / subdivide / to_dense) -> marching cubes, on a
    ragged scene, with every wrapper-allocated output guarded."""
    from rfdnet_amd.iscnet.config import Config
    from rfdnet_amd.iscnet.network import ISCNet
    cfg = Config({'data': {'num_point': 20000}, 'generation': {'resolution_0': 16 if steps else 20, 'upsampling_steps': steps}})
    net = ISCNet(cfg)
    synthetic.load_seeded(net, 10)
    net = net.cuda().eval()
    pc = torch.from_numpy(synthetic.synthetic_scene(seed=6, n_raw=30000, n_points=20000)[None]).cuda()
    g = Guards()
    with g.active(), torch.no_grad():
        ep, pf = net.detect(pc)
        ids = net.select_proposals(ep, 'all', pc)[:, :37].contiguous()           # 37 proposals: ragged tile counts
        meshes = net.reconstruct(ep, pf, ids, pc)
    n, bad = g.check()
    hip.device_status()
    assert len(meshes) == 37 and n >= 20, n
    assert bad == [], bad


@pytest.mark.parametrize("kern", ["w8", "w4"])
def test_decoder_logits_stay_inside_their_buffer(hip, kern):
    from rfdnet_amd.iscnet.occ_decoder import DecoderCBatchNorm
    dec = DecoderCBatchNorm(dim=3, z_dim=32, c_dim=512, hidden_size=256)
    synthetic.load_seeded(dec, 3)
    dec = dec.cuda().eval()
    dec.kernel = kern
    rng = np.random.default_rng(1)
    p = torch.from_numpy(((rng.random((3, 333, 3)) - 0.5) * 1.1).astype(np.float32)).cuda()
    z = torch.zeros(3, 32).cuda()
    c = torch.from_numpy(rng.normal(0, 1, (3, 512)).astype(np.float32)).cuda()
    g = Guards()
    with g.active(), torch.no_grad():
        out = dec(p, z, c)
    n, bad = g.check()
    hip.device_status()
    assert out.shape == (3, 333) and n >= 2
    assert bad == [], bad
