"""Deterministic, torch-version-independent synthetic weights and scenes
(stand-ins for the absent pretrained weights and ScanNet scans).

torch's RNG is not stable across versions/devices, so every synthetic tensor
used by fixtures, tests and bench comes from numpy's PCG64 (`default_rng`),
which is specified bit-for-bit.  Shared by tests/golden/make_fixtures.py (dev
container, with the reference importable) and the tests themselves (anywhere).
"""
from collections import OrderedDict

import numpy as np


def seeded_tensor(name, shape, rng):
    """Value rule by parameter name (BN running_mean~N(0,0.1),
    running_var~U(0.5,1.5), conv weights uniform(+-1/sqrt(fan_in)), CBN
    gamma/beta convs N(0,0.02) so the conditioning path is exercised -- the
    reference zero-initialises them, layers.py:220-224)."""
    shape = tuple(int(s) for s in shape)
    if name.endswith("num_batches_tracked"):
        return np.zeros(shape, dtype=np.int64)
    if name.endswith("running_var"):
        return rng.uniform(0.5, 1.5, shape).astype(np.float32)
    if name.endswith("running_mean"):
        return rng.normal(0.0, 0.1, shape).astype(np.float32)
    if "conv_gamma.weight" in name or "conv_beta.weight" in name:
        return rng.normal(0.0, 0.02, shape).astype(np.float32)
    if "conv_gamma.bias" in name:
        return (1.0 + rng.normal(0.0, 0.1, shape)).astype(np.float32)
    if name.endswith(".weight") and len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        b = 1.0 / np.sqrt(fan_in)
        return rng.uniform(-b, b, shape).astype(np.float32)
    if name.endswith(".weight"):          # affine BN scale
        return rng.uniform(0.5, 1.5, shape).astype(np.float32)
    if name.endswith(".bias"):
        return rng.uniform(-0.1, 0.1, shape).astype(np.float32)
    raise KeyError("no seeding rule for %s" % name)


def seeded_state_dict(shapes, seed):
    """shapes: ordered {name: shape}.  Returns ordered {name: ndarray}."""
    rng = np.random.default_rng(seed)
    out = OrderedDict()
    for k, s in shapes.items():
        out[k] = seeded_tensor(k, s, rng)
    return out


def load_seeded(module, seed):
    """Overwrite every parameter/buffer of a torch module from the seed.
    Returns the ordered {name: shape} map (for key-parity checks)."""
    import torch
    sd = module.state_dict()
    shapes = OrderedDict((k, tuple(v.shape)) for k, v in sd.items())
    new = seeded_state_dict(shapes, seed)
    module.load_state_dict({k: torch.from_numpy(v) for k, v in new.items()})
    return shapes


def synthetic_scene(seed=10, n_raw=120000, n_points=80000, with_origin_pts=True):
    """ScanNet-like room: 6x7x2.8 m, floor + 4 walls + 12
    cuboid 'furniture' surfaces, sigma=5 mm noise, 16 points within 0.02 m of
    the origin (exercise the FPS |p|^2 <= 1e-3 skip), height channel =
    z - percentile(z, 0.99) (demo.py:38-40), random subsample to n_points
    (with replacement when n_raw < n_points, pc_util.py:35-47).
    Returns (n_points, 4) float32."""
    rng = np.random.default_rng(seed)
    W, L, Hh = 6.0, 7.0, 2.8
    parts = []

    def plane(n, o, u, v):
        a = rng.random((n, 1))
        b = rng.random((n, 1))
        return np.asarray(o)[None] + a * np.asarray(u)[None] + b * np.asarray(v)[None]

    n_floor = int(n_raw * 0.30)
    n_wall = int(n_raw * 0.08)
    parts.append(plane(n_floor, (-W / 2, -L / 2, 0), (W, 0, 0), (0, L, 0)))
    parts.append(plane(n_wall, (-W / 2, -L / 2, 0), (W, 0, 0), (0, 0, Hh)))
    parts.append(plane(n_wall, (-W / 2, L / 2, 0), (W, 0, 0), (0, 0, Hh)))
    parts.append(plane(n_wall, (-W / 2, -L / 2, 0), (0, L, 0), (0, 0, Hh)))
    parts.append(plane(n_wall, (W / 2, -L / 2, 0), (0, L, 0), (0, 0, Hh)))
    n_left = n_raw - n_floor - 4 * n_wall - (16 if with_origin_pts else 0)
    n_obj = 12
    per = n_left // n_obj
    for i in range(n_obj):
        size = rng.uniform(0.4, 1.6, 3) * np.array([1.0, 1.0, 0.7])
        ctr = np.array([rng.uniform(-W / 2 + 0.8, W / 2 - 0.8),
                        rng.uniform(-L / 2 + 0.8, L / 2 - 0.8), size[2] / 2])
        ang = rng.uniform(0, np.pi)
        n_i = per if i < n_obj - 1 else n_left - per * (n_obj - 1)
        face = rng.integers(0, 5, n_i)          # 4 sides + top
        uv = rng.random((n_i, 2)) - 0.5
        p = np.zeros((n_i, 3))
        for f in range(5):
            m = face == f
            if f == 4:
                p[m] = np.stack([uv[m, 0] * size[0], uv[m, 1] * size[1],
                                 np.full(m.sum(), size[2] / 2)], 1)
            elif f < 2:
                p[m] = np.stack([np.full(m.sum(), (f - 0.5) * size[0]),
                                 uv[m, 0] * size[1], uv[m, 1] * size[2]], 1)
            else:
                p[m] = np.stack([uv[m, 0] * size[0],
                                 np.full(m.sum(), (f - 2.5) * size[1]),
                                 uv[m, 1] * size[2]], 1)
        c, s = np.cos(ang), np.sin(ang)
        R = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])
        parts.append(p @ R.T + ctr[None])
    pts = np.concatenate(parts, 0)
    pts = pts + rng.normal(0, 0.005, pts.shape)
    if with_origin_pts:
        pts = np.concatenate([pts, rng.uniform(-0.0115, 0.0115, (16, 3))], 0)
    pts = pts.astype(np.float32)
    floor_h = np.percentile(pts[:, 2], 0.99)
    height = (pts[:, 2] - floor_h).astype(np.float32)
    pc = np.concatenate([pts, height[:, None]], 1).astype(np.float32)
    replace = pc.shape[0] < n_points
    choice = rng.choice(pc.shape[0], n_points, replace=replace)
    return np.ascontiguousarray(pc[choice])
