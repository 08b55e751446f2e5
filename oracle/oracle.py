"""numpy/ctypes front-end of the CPU oracle (oracle/rfd_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never from rfdnet_amd/ (the product).

All functions take and return numpy arrays with the reference's layouts
(see each C function's header comment for the reference file:line).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# RFD_ORACLE_LIB: another build of the same source (the sanitizer build, tests/test_oracle_sanitized.py)
_LIB_PATH = os.environ.get("RFD_ORACLE_LIB") or os.path.join(_HERE, "liboracle.so")
_lib = None

_f32p = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int)
_i64p = C.POINTER(C.c_int64)
_f64p = C.POINTER(C.c_double)


def build(force=False):
    """Compile liboracle.so with gcc (oracle/Makefile)."""
    deps = [os.path.join(_HERE, "rfd_oracle.c"),
            os.path.join(os.path.dirname(_HERE), "rfdnet_amd", "csrc", "mc_tables.h")]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(d) for d in deps)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", os.path.basename(_LIB_PATH)],
                          stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.oracle_mise_create.restype = C.c_void_p
        _lib.oracle_mise_create.argtypes = [C.c_int, C.c_int, C.c_double]
        _lib.oracle_mise_destroy.argtypes = [C.c_void_p]
        _lib.oracle_mise_resolution.argtypes = [C.c_void_p]
        _lib.oracle_mise_query.restype = C.c_long
        _lib.oracle_mise_query.argtypes = [C.c_void_p, _i64p, C.c_long]
        _lib.oracle_mise_update.argtypes = [C.c_void_p, _i64p, _f64p, C.c_long]
        _lib.oracle_mise_to_dense.argtypes = [C.c_void_p, _f64p]
        _lib.oracle_make_3d_grid.argtypes = [C.c_float, C.c_float, C.c_int,
                                             C.c_float, _f32p]
        _lib.oracle_ball_query.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float,
                                           C.c_int, _f32p, _f32p, _i32p]
    return _lib


class variant(object):
    """`with oracle.variant(1): ...` -- run the point ops with another a*a+b*b+c*c evaluation
    order (rfd_oracle.c ORACLE_SUMSQ_VARIANT: 1 = right-nested fma, 2 = no contraction).  Used only
    to measure how much depends on the nvcc fma-order assumption."""

    def __init__(self, v):
        self.v = int(v)

    def __enter__(self):
        global _lib
        lib()
        self.saved = _lib
        if self.v:
            path = os.path.join(_HERE, "liboracle_v%d.so" % self.v)
            src = os.path.join(_HERE, "rfd_oracle.c")
            if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
                subprocess.check_call(["make", "-C", _HERE, "-B", os.path.basename(path)],
                                      stdout=subprocess.DEVNULL)
            l = C.CDLL(path)
            assert l.oracle_sumsq_variant() == self.v
            l.oracle_ball_query.argtypes = self.saved.oracle_ball_query.argtypes
            _lib = l
        return self

    def __exit__(self, *a):
        global _lib
        _lib = self.saved


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_f32p)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_i32p)


def num_threads():
    return lib().oracle_num_threads()


def set_num_threads(n):
    lib().oracle_set_num_threads(int(n))


def opt_n_threads(n):
    return lib().oracle_opt_n_threads(int(n))


# ---- point ops (the 9 functions of bindings.cpp:6-19) -----------------------

def furthest_point_sampling(points, nsamples, return_temp=False):
    """points (B,N,3) f32 -> idx (B,M) i32.  sampling.cpp:66-87."""
    points, pp = _f(points)
    b, n, _ = points.shape
    out = np.zeros((b, nsamples), dtype=np.int32)          # sampling.cpp:70-72
    tmp = np.full((b, n), 1e10, dtype=np.float32)          # sampling.cpp:74-76
    lib().oracle_furthest_point_sampling(b, n, nsamples, pp,
                                         tmp.ctypes.data_as(_f32p),
                                         out.ctypes.data_as(_i32p))
    return (out, tmp) if return_temp else out


def gather_points(points, idx):
    """points (B,C,N), idx (B,M) -> (B,C,M).  sampling.cpp:15-38."""
    points, pp = _f(points)
    idx, ip = _i(idx)
    b, c, n = points.shape
    m = idx.shape[1]
    out = np.zeros((b, c, m), dtype=np.float32)
    lib().oracle_gather_points(b, c, n, m, pp, ip, out.ctypes.data_as(_f32p))
    return out


def gather_points_grad(grad_out, idx, n):
    grad_out, gp = _f(grad_out)
    idx, ip = _i(idx)
    b, c, m = grad_out.shape
    out = np.zeros((b, c, n), dtype=np.float32)
    lib().oracle_gather_points_grad(b, c, n, m, gp, ip, out.ctypes.data_as(_f32p))
    return out


def ball_query(new_xyz, xyz, radius, nsample):
    """new_xyz (B,M,3), xyz (B,N,3) -> idx (B,M,ns) i32.  ball_query.cpp:8-32."""
    new_xyz, cp = _f(new_xyz)
    xyz, xp = _f(xyz)
    b, n, _ = xyz.shape
    m = new_xyz.shape[1]
    idx = np.zeros((b, m, nsample), dtype=np.int32)        # ball_query.cpp:19-21
    lib().oracle_ball_query(b, n, m, C.c_float(radius), nsample, cp, xp,
                            idx.ctypes.data_as(_i32p))
    return idx


def group_points(points, idx):
    """points (B,C,N), idx (B,M,ns) -> (B,C,M,ns).  group_points.cpp:12-36."""
    points, pp = _f(points)
    idx, ip = _i(idx)
    b, c, n = points.shape
    _, m, ns = idx.shape
    out = np.zeros((b, c, m, ns), dtype=np.float32)
    lib().oracle_group_points(b, c, n, m, ns, pp, ip, out.ctypes.data_as(_f32p))
    return out


def group_points_grad(grad_out, idx, n):
    grad_out, gp = _f(grad_out)
    idx, ip = _i(idx)
    b, c, m, ns = grad_out.shape
    out = np.zeros((b, c, n), dtype=np.float32)
    lib().oracle_group_points_grad(b, c, n, m, ns, gp, ip, out.ctypes.data_as(_f32p))
    return out


def three_nn(unknown, known):
    """unknown (B,n,3), known (B,m,3) -> dist2 (B,n,3) f32, idx (B,n,3) i32."""
    unknown, up = _f(unknown)
    known, kp = _f(known)
    b, n, _ = unknown.shape
    m = known.shape[1]
    d2 = np.zeros((b, n, 3), dtype=np.float32)
    idx = np.zeros((b, n, 3), dtype=np.int32)
    lib().oracle_three_nn(b, n, m, up, kp, d2.ctypes.data_as(_f32p),
                          idx.ctypes.data_as(_i32p))
    return d2, idx


def chamfer_forward(xyz1, xyz2):
    """xyz1 (B,n,3), xyz2 (B,m,3) -> dist1 (B,n) f32, idx1 (B,n) i32, dist2 (B,m), idx2 (B,m)
    (chamfer_distance.cpp:60-115)."""
    xyz1, p1 = _f(xyz1)
    xyz2, p2 = _f(xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    d1 = np.zeros((b, n), dtype=np.float32)
    i1 = np.zeros((b, n), dtype=np.int32)
    d2 = np.zeros((b, m), dtype=np.float32)
    i2 = np.zeros((b, m), dtype=np.int32)
    lib().oracle_chamfer_forward(b, n, m, p1, p2, d1.ctypes.data_as(_f32p), i1.ctypes.data_as(_i32p),
                                 d2.ctypes.data_as(_f32p), i2.ctypes.data_as(_i32p))
    return d1, i1, d2, i2


def chamfer_backward(xyz1, xyz2, gd1, idx1, gd2, idx2):
    """-> grad_xyz1 (B,n,3), grad_xyz2 (B,m,3) (chamfer_distance.cpp:118-180)."""
    xyz1, p1 = _f(xyz1)
    xyz2, p2 = _f(xyz2)
    gd1, q1 = _f(gd1)
    gd2, q2 = _f(gd2)
    idx1, j1 = _i(idx1)
    idx2, j2 = _i(idx2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    g1 = np.zeros((b, n, 3), dtype=np.float32)
    g2 = np.zeros((b, m, 3), dtype=np.float32)
    lib().oracle_chamfer_backward(b, n, m, p1, p2, q1, j1, q2, j2, g1.ctypes.data_as(_f32p),
                                  g2.ctypes.data_as(_f32p))
    return g1, g2


def three_interpolate(points, idx, weight):
    """points (B,C,m), idx/weight (B,n,3) -> (B,C,n)."""
    points, pp = _f(points)
    idx, ip = _i(idx)
    weight, wp = _f(weight)
    b, c, m = points.shape
    n = idx.shape[1]
    out = np.zeros((b, c, n), dtype=np.float32)
    lib().oracle_three_interpolate(b, c, m, n, pp, ip, wp, out.ctypes.data_as(_f32p))
    return out


def three_interpolate_grad(grad_out, idx, weight, m):
    grad_out, gp = _f(grad_out)
    idx, ip = _i(idx)
    weight, wp = _f(weight)
    b, c, n = grad_out.shape
    out = np.zeros((b, c, m), dtype=np.float32)
    lib().oracle_three_interpolate_grad(b, c, n, m, gp, ip, wp, out.ctypes.data_as(_f32p))
    return out


# ---- occupancy decoder -------------------------------------------------------

def decoder_param_blob(sd, prefix=""):
    """Flatten a DecoderCBatchNorm state_dict (reference key names,
    occ_decoder.py:85-108) into the oracle's parameter blob."""
    def g(k):
        v = sd[prefix + k]
        v = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
        return np.ascontiguousarray(v, dtype=np.float32).reshape(-1)

    def cbn(p):
        return [g(p + ".conv_gamma.weight"), g(p + ".conv_gamma.bias"),
                g(p + ".conv_beta.weight"), g(p + ".conv_beta.bias"),
                g(p + ".bn.running_mean"), g(p + ".bn.running_var")]

    parts = [g("fc_p.weight"), g("fc_p.bias"), g("fc_z.weight"), g("fc_z.bias")]
    for i in range(5):
        parts += cbn("blocks.%d.bn_0" % i)
        parts += [g("blocks.%d.fc_0.weight" % i), g("blocks.%d.fc_0.bias" % i)]
        parts += cbn("blocks.%d.bn_1" % i)
        parts += [g("blocks.%d.fc_1.weight" % i), g("blocks.%d.fc_1.bias" % i)]
    parts += cbn("bn")
    parts += [g("fc_out.weight"), g("fc_out.bias")]
    return np.concatenate(parts)


def decoder_cbn(blob, p, z, c, hidden=256):
    """p (K,T,3), z (K,Z), c (K,C) -> logits (K,T).  occ_decoder.py:110-123."""
    p, pp = _f(p)
    z, zp = _f(z)
    c, cp = _f(c)
    blob, bp = _f(blob)
    K, T, _ = p.shape
    Z, Cd = z.shape[1], c.shape[1]
    H = hidden
    expect = H * 3 + H + H * Z + H + 11 * (2 * H * Cd + 4 * H) + 10 * (H * H + H) + H + 1
    assert blob.size == expect, (blob.size, expect)
    out = np.zeros((K, T), dtype=np.float32)
    lib().oracle_decoder_cbn(K, T, H, Cd, Z, bp, pp, zp, cp, out.ctypes.data_as(_f32p))
    return out


def make_3d_grid(lo, hi, n, scale=1.0):
    """external/common.py:157-176 with the generator's box_size scale folded."""
    out = np.zeros((n * n * n, 3), dtype=np.float32)
    lib().oracle_make_3d_grid(C.c_float(lo), C.c_float(hi), n, C.c_float(scale),
                              out.ctypes.data_as(_f32p))
    return out


# ---- MISE --------------------------------------------------------------------

class MISE(object):
    """Same public surface as external/libmise/mise.pyx:33 (query/update/to_dense)."""

    def __init__(self, resolution_0, depth, threshold):
        self._h = lib().oracle_mise_create(int(resolution_0), int(depth), float(threshold))
        self.resolution_0 = resolution_0
        self.depth = depth
        self.threshold = threshold
        self.resolution = lib().oracle_mise_resolution(self._h)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().oracle_mise_destroy(self._h)
            self._h = None

    def query(self):
        n = lib().oracle_mise_query(self._h, None, 0)
        out = np.zeros((n, 3), dtype=np.int64)
        if n:
            lib().oracle_mise_query(self._h, out.ctypes.data_as(_i64p), n)
        return out

    def update(self, points, values):
        points = np.ascontiguousarray(points, dtype=np.int64)
        values = np.ascontiguousarray(values, dtype=np.float64)
        assert points.shape[0] == values.shape[0] and points.shape[1] == 3
        rc = lib().oracle_mise_update(self._h, points.ctypes.data_as(_i64p),
                                      values.ctypes.data_as(_f64p), points.shape[0])
        if rc != 0:
            raise ValueError("Point not in grid!")

    def to_dense(self):
        r1 = self.resolution + 1
        out = np.zeros((r1, r1, r1), dtype=np.float64)
        lib().oracle_mise_to_dense(self._h, out.ctypes.data_as(_f64p))
        return out


# ---- a `pointnet2_ops._ext` look-alike over torch CPU tensors -----------------
# Used ONLY by tests/golden/make_fixtures.py to run the reference's own Python
# modules on CPU in the dev container.

class TorchExt(object):
    """The 9 functions of bindings.cpp:6-19 on torch CPU tensors."""

    @staticmethod
    def _t(a):
        import torch
        return torch.from_numpy(a)

    def gather_points(self, points, idx):
        return self._t(gather_points(points.numpy(), idx.numpy()))

    def gather_points_grad(self, grad_out, idx, n):
        return self._t(gather_points_grad(grad_out.numpy(), idx.numpy(), n))

    def furthest_point_sampling(self, points, nsamples):
        return self._t(furthest_point_sampling(points.numpy(), nsamples))

    def three_nn(self, unknowns, knows):
        d2, idx = three_nn(unknowns.numpy(), knows.numpy())
        return self._t(d2), self._t(idx)

    def three_interpolate(self, points, idx, weight):
        return self._t(three_interpolate(points.numpy(), idx.numpy(), weight.numpy()))

    def three_interpolate_grad(self, grad_out, idx, weight, m):
        return self._t(three_interpolate_grad(grad_out.numpy(), idx.numpy(), weight.numpy(), m))

    def ball_query(self, new_xyz, xyz, radius, nsample):
        return self._t(ball_query(new_xyz.numpy(), xyz.numpy(), radius, nsample))

    def group_points(self, points, idx):
        return self._t(group_points(points.numpy(), idx.numpy()))

    def group_points_grad(self, grad_out, idx, n):
        return self._t(group_points_grad(grad_out.numpy(), idx.numpy(), n))


# ---- marching cubes (PyMCubes' algorithm restated; see rfd_oracle.c) --------------
def marching_cubes(volume, isovalue):
    """volume (nx,ny,nz) -> (vertices (nv,3) f64 in index coordinates, triangles (nt,3) i32),
    with the library's vertex / triangle order.  The CALLER pads (generator.py:158-159)."""
    g = np.ascontiguousarray(volume, dtype=np.float64)
    nx, ny, nz = g.shape
    fn = lib().oracle_marching_cubes
    fn.argtypes = [C.c_int, C.c_int, C.c_int, _f64p, C.c_double, _f64p, C.POINTER(C.c_long),
                   _i32p, C.POINTER(C.c_long)]
    nv, nt = C.c_long(0), C.c_long(0)
    gp = g.ctypes.data_as(_f64p)
    rc = fn(nx, ny, nz, gp, float(isovalue), None, C.byref(nv), None, C.byref(nt))
    assert rc == 0, "oracle_marching_cubes: %d" % rc
    v = np.empty((nv.value, 3), np.float64)
    t = np.empty((nt.value, 3), np.int32)
    rc = fn(nx, ny, nz, gp, float(isovalue), v.ctypes.data_as(_f64p), C.byref(nv),
            t.ctypes.data_as(_i32p), C.byref(nt))
    assert rc == 0
    return v, t


def extract_mesh(occ_hat, threshold_logit, padding=0.1):
    """Generator3D.extract_mesh (generator.py:145-168) on one value grid: pad with -1e6,
    marching cubes, `-0.5`, `-1`, `/(n-1)`, `box*(v-0.5)`."""
    occ_hat = np.asarray(occ_hat, dtype=np.float64)
    n = np.array(occ_hat.shape, dtype=np.float64)
    padded = np.pad(occ_hat, 1, 'constant', constant_values=-1e6)
    v, t = marching_cubes(padded, threshold_logit)
    v = v - 0.5
    v = v - 1
    v = v / (n - 1)
    v = (1 + padding) * (v - 0.5)
    return v, t
