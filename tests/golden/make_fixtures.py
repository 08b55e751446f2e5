"""Generate the golden fixtures under tests/golden/ (DEV CONTAINER ONLY).

Imports the reference's own Python from /root/reference (never shipped), runs
it on CPU and stores inputs/outputs as small .npz files.  Weights are NOT
stored: both this script and the tests regenerate them from numpy PCG64 seeds
(rfdnet_amd/synthetic.py), the fixtures keep only the parameter name/shape
lists (which also pin state_dict key parity), inputs and reference outputs.

Recipe = SURVEY.md Appendix A: bare namespace packages so models/__init__.py
(-> loss.py -> chamfer JIT + .cuda()) never executes, `pointnet2_ops._ext`
replaced by the CPU oracle, absent third-party deps stubbed, MISE built from
the reference's mise.pyx with Cython in a scratch directory.

Usage:  python tests/golden/make_fixtures.py [dec] [mise] [grid] [ops] [net] [gen]
"""
import importlib
import os
import subprocess
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from oracle import oracle  # noqa: E402
from rfdnet_amd import synthetic  # noqa: E402


def ns(name, path=None):
    m = types.ModuleType(name)
    m.__path__ = [path] if path else []
    sys.modules[name] = m
    return m


def mount_reference():
    import torch  # noqa: F401
    sys.path.insert(0, REF)
    ns('models', REF + '/models')
    ns('models.iscnet', REF + '/models/iscnet')
    ns('models.iscnet.modules', REF + '/models/iscnet/modules')
    ext = oracle.TorchExt()
    pkg = ns('pointnet2_ops', REF + '/external/pointnet2_ops_lib/pointnet2_ops')
    sys.modules['pointnet2_ops._ext'] = ext
    pkg._ext = ext
    ns('trimesh').Trimesh = lambda *a, **k: a
    ns('mcubes').marching_cubes = lambda vol, thr: (np.zeros((0, 3)), np.zeros((0, 3), int))
    ns('external', REF + '/external')
    ns('external.libsimplify').simplify_mesh = None
    ns('external.libkdtree')
    ns('external.libkdtree.pykdtree')
    ns('external.libkdtree.pykdtree.kdtree').KDTree = None


def build_ref_mise():
    """cythonize the reference's mise.pyx where it lies; output to a scratch dir."""
    d = tempfile.mkdtemp(prefix="refmise_")
    setup = os.path.join(d, "setup.py")
    with open(setup, "w") as f:
        f.write("from setuptools import setup, Extension\n"
                "from Cython.Build import cythonize\nimport numpy\n"
                "setup(ext_modules=cythonize([Extension('mise', ['%s/external/libmise/mise.pyx'],"
                " language='c++', include_dirs=[numpy.get_include()])], build_dir='%s/build',"
                " language_level=3))\n" % (REF, d))
    subprocess.check_call([sys.executable, setup, "build_ext", "--build-lib", d, "--build-temp",
                           d + "/tmp"], cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    sys.path.insert(0, d)
    return importlib.import_module("mise")


def shapes_arrays(shapes):
    names = np.array(list(shapes.keys()))
    shp = np.array([",".join(str(int(x)) for x in s) for s in shapes.values()])
    return names, shp


# ------------------------------------------------------------------ F-DEC ----
def make_dec():
    import torch
    mount_reference()
    occ = importlib.import_module('models.iscnet.modules.occ_decoder')
    torch.manual_seed(0)
    dec = occ.DecoderCBatchNorm(dim=3, z_dim=32, c_dim=512, hidden_size=256)
    shapes = synthetic.load_seeded(dec, seed=1234)
    dec.eval()
    rng = np.random.default_rng(77)
    K, T = 3, 1536
    p = ((rng.random((K, T, 3)) - 0.5) * 1.1).astype(np.float32)
    z = np.zeros((K, 32), dtype=np.float32)
    z[2] = rng.normal(0, 1, 32).astype(np.float32)          # one non-prior z
    c = rng.normal(0, 1, (K, 512)).astype(np.float32)
    with torch.no_grad():
        out = dec(torch.from_numpy(p), torch.from_numpy(z), torch.from_numpy(c)).numpy()
    names, shp = shapes_arrays(shapes)
    np.savez_compressed(os.path.join(HERE, "F_DEC.npz"), seed=1234, names=names, shapes=shp,
                        p=p, z=z, c=c, logits=out.astype(np.float32))
    print("F_DEC: logits", out.shape, float(out.min()), float(out.max()), float(out.std()))


# ----------------------------------------------------------------- F-MISE ----
def sphere_field(pts, res, r=0.35):
    q = pts.astype(np.float64) / res - 0.5
    return r - np.sqrt((q ** 2).sum(-1))


def run_mise(MISE, res0, depth, thr, field):
    m = MISE(res0, depth, thr)
    rounds = []
    p = m.query()
    while p.shape[0] != 0 and len(rounds) < 16:
        v = field(p, m.resolution)
        rounds.append((p.copy(), v.copy()))
        m.update(p, v)
        p = m.query()
    return rounds, m.to_dense()


def make_mise():
    mise = build_ref_mise()
    out = {}
    cases = {
        "sphere_8_2": (8, 2, 0.0, lambda p, r: sphere_field(p, r)),
        "sphere_16_1": (16, 1, 0.0, lambda p, r: sphere_field(p, r)),
        # external/libmise/test.py: MISE(1,2,0.), v = 2*(sum p > 2) - 1
        "libmise_test": (1, 2, 0.0, lambda p, r: 2 * (p.sum(axis=-1) > 2).astype(np.float64) - 1),
        # values exactly at the threshold exercise the non-strict >= / <= (mise.pyx:225-227)
        "plane_eq_4_2": (4, 2, 0.0, lambda p, r: (p[:, 0] - r // 2).astype(np.float64)),
        "empty_4_1": (4, 1, 0.0, lambda p, r: -np.ones(p.shape[0])),
    }
    for name, (res0, depth, thr, field) in cases.items():
        rounds, dense = run_mise(mise.MISE, res0, depth, thr, field)
        out[name + "_cfg"] = np.array([res0, depth, thr], dtype=np.float64)
        out[name + "_nrounds"] = np.array(len(rounds))
        for i, (p, v) in enumerate(rounds):
            out["%s_p%d" % (name, i)] = p.astype(np.int16)
            out["%s_v%d" % (name, i)] = v
        out[name + "_dense"] = dense
        print("F_MISE", name, [r[0].shape[0] for r in rounds], dense.shape)
    # headline shape: only the per-round counts + a checksum (the lists are large)
    rounds, dense = run_mise(mise.MISE, 32, 1, 0.0, lambda p, r: sphere_field(p, r))
    out["sphere_32_1_counts"] = np.array([r[0].shape[0] for r in rounds])
    out["sphere_32_1_dense_sum"] = np.array([dense.sum(), np.abs(dense).sum(), (dense > 0).sum()])
    print("F_MISE sphere_32_1", out["sphere_32_1_counts"])
    np.savez_compressed(os.path.join(HERE, "F_MISE.npz"), **out)


# ----------------------------------------------------------------- F-GRID ----
def make_grid():
    import torch
    mount_reference()
    common = importlib.import_module('external.common')
    out = {}
    for nx in (2, 3, 8, 32, 33):
        g = 1.1 * common.make_3d_grid((-0.5,) * 3, (0.5,) * 3, (nx,) * 3)   # generator.py:92-94
        out["grid_%d" % nx] = g.numpy() if nx <= 8 else g.numpy()[:: max(1, nx * nx * nx // 4096)]
        out["axis_%d" % nx] = (1.1 * torch.linspace(-0.5, 0.5, nx)).numpy()
    np.savez_compressed(os.path.join(HERE, "F_GRID.npz"), **out)
    print("F_GRID ok")


if __name__ == "__main__":
    what = sys.argv[1:] or ["dec", "mise", "grid"]
    for w in what:
        globals()["make_" + w]()
