"""CPU: the C oracle of the six point ops against an INDEPENDENT numpy
restatement and against defining properties (the reference holds no golden
vectors for these ops -- SURVEY.md §4 / §8c)."""
import numpy as np
import pytest


def f32(x):
    return np.asarray(x, dtype=np.float32)


def sumsq3(a, b, c):
    """contract order t=b*b; fma(a,a,t); fma(c,c,t) emulated in float64 with
    explicit float32 roundings (exact: a float32 product fits in float64)."""
    a, b, c = (np.asarray(v, dtype=np.float32).astype(np.float64) for v in (a, b, c))
    t = f32(b * b).astype(np.float64)
    t = f32(a * a + t).astype(np.float64)
    return f32(c * c + t)


def np_fps(xyz, m):
    """Semantics of sampling_gpu.cu:69-173 via the closed-form tie rule."""
    n = xyz.shape[0]
    bs = min(1 << int(np.log(float(n)) / np.log(2.0)), 512)
    lg = int(np.log2(bs))
    k = np.arange(n)
    tid = k % bs
    rev = np.array([int(format(t, "0%db" % lg)[::-1], 2) if lg else 0 for t in range(bs)])[tid]
    rank = rev.astype(np.int64) * ((n + bs - 1) // bs) + k // bs
    mag = sumsq3(xyz[:, 0], xyz[:, 1], xyz[:, 2])
    valid = ~(mag.astype(np.float64) <= 1e-3)
    temp = np.full(n, 1e10, dtype=np.float32)
    idx = np.zeros(m, dtype=np.int32)
    old = 0
    for j in range(1, m):
        d = sumsq3(xyz[:, 0] - xyz[old, 0], xyz[:, 1] - xyz[old, 1], xyz[:, 2] - xyz[old, 2])
        temp = np.where(valid, np.minimum(d, temp), temp)
        if not valid.any():
            old = 0
        else:
            best = temp[valid].max()
            cand = np.where(valid & (temp == best))[0]
            old = int(cand[np.argmin(rank[cand])])
        idx[j] = old
    return idx, temp


def np_ball_query(new_xyz, xyz, radius, ns):
    r2 = np.float32(radius) * np.float32(radius)
    out = np.zeros((new_xyz.shape[0], ns), dtype=np.int32)
    for j, c in enumerate(new_xyz):
        d2 = sumsq3(c[0] - xyz[:, 0], c[1] - xyz[:, 1], c[2] - xyz[:, 2])
        hits = np.where(d2 < r2)[0][:ns]
        if hits.size:
            out[j, :] = hits[0]
            out[j, :hits.size] = hits
    return out


def scene(rng, n, dup=0, origin=0):
    p = f32(rng.uniform(-2, 2, (n, 3)))
    if dup:
        src = rng.integers(0, n, dup)
        dst = rng.integers(0, n, dup)
        p[dst] = p[src]                      # exact duplicates (random_sampling with replacement)
    if origin:
        p[rng.integers(0, n, origin)] = f32(rng.uniform(-0.01, 0.01, (origin, 3)))
    return p


@pytest.mark.parametrize("n,m,dup,origin", [(37, 11, 0, 0), (512, 64, 40, 5), (700, 128, 100, 8),
                                            (1500, 200, 300, 16), (64, 64, 0, 0)])
def test_fps_matches_numpy(oracle, n, m, dup, origin):
    rng = np.random.default_rng(n * 7 + m)
    p = scene(rng, n, dup, origin)
    idx, temp = oracle.furthest_point_sampling(p[None], m, return_temp=True)
    ridx, rtemp = np_fps(p, m)
    np.testing.assert_array_equal(idx[0], ridx)
    np.testing.assert_array_equal(temp[0], rtemp)


def test_fps_ties_on_lattice(oracle):
    """integer lattice => massive exact distance ties: the CUDA tree order decides."""
    g = np.stack(np.meshgrid(*[np.arange(9)] * 3, indexing="ij"), -1).reshape(-1, 3)
    rng = np.random.default_rng(5)
    p = f32(g[rng.permutation(g.shape[0])] + 1.0)
    idx = oracle.furthest_point_sampling(p[None], 100)
    ridx, _ = np_fps(p, 100)
    np.testing.assert_array_equal(idx[0], ridx)


def test_fps_all_points_skipped(oracle):
    p = f32(np.random.default_rng(0).uniform(-0.01, 0.01, (100, 3)))
    idx = oracle.furthest_point_sampling(p[None], 10)
    assert (idx == 0).all()                   # every |p|^2 <= 1e-3 -> index 0 (sampling_gpu.cu:90,101)


def test_fps_property_greedy(oracle):
    """each pick maximises the min distance to the previous picks"""
    rng = np.random.default_rng(3)
    p = f32(rng.uniform(1, 3, (400, 3)))
    idx = oracle.furthest_point_sampling(p[None], 50)[0]
    assert len(set(idx.tolist())) == 50
    for j in range(1, 50):
        d = ((p[:, None, :].astype(np.float64) - p[idx[:j]][None].astype(np.float64)) ** 2).sum(-1).min(1)
        assert d[idx[j]] >= d.max() * (1 - 1e-5)


@pytest.mark.parametrize("n,m,ns,r", [(300, 20, 8, 0.5), (1000, 64, 16, 0.3), (257, 33, 64, 4.0),
                                      (128, 16, 4, 0.01)])
def test_ball_query_matches_numpy(oracle, n, m, ns, r):
    rng = np.random.default_rng(n + m)
    xyz = scene(rng, n, dup=n // 10)
    new = np.concatenate([xyz[rng.integers(0, n, m - 2)], f32([[50, 50, 50], [-50, 0, 0]])])
    idx = oracle.ball_query(new[None], xyz[None], r, ns)
    ref = np_ball_query(new, xyz, r, ns)
    np.testing.assert_array_equal(idx[0], ref)
    assert (idx[0, -1] == 0).all() and (idx[0, -2] == 0).all()     # empty balls stay zero


def test_ball_query_boundary_is_strict(oracle):
    xyz = f32([[0, 0, 0], [1, 0, 0], [0.5, 0, 0]])
    idx = oracle.ball_query(f32([[0, 0, 0]])[None], xyz[None], 1.0, 3)
    np.testing.assert_array_equal(idx[0, 0], [0, 2, 0])             # d2 == r^2 excluded (:33)


def test_group_gather_interpolate(oracle):
    rng = np.random.default_rng(9)
    B, C, N, M, ns = 2, 5, 40, 7, 3
    pts = f32(rng.normal(size=(B, C, N)))
    idx = rng.integers(0, N, (B, M, ns)).astype(np.int32)
    out = oracle.group_points(pts, idx)
    ref = np.stack([pts[b][:, idx[b]] for b in range(B)])
    np.testing.assert_array_equal(out, ref)
    gi = rng.integers(0, N, (B, M)).astype(np.int32)
    np.testing.assert_array_equal(oracle.gather_points(pts, gi),
                                  np.stack([pts[b][:, gi[b]] for b in range(B)]))
    w = f32(rng.random((B, M, 3)))
    i3 = rng.integers(0, N, (B, M, 3)).astype(np.int32)
    o = oracle.three_interpolate(pts, i3, w)
    ref = np.zeros((B, C, M), dtype=np.float32)
    for b in range(B):
        p = pts[b].astype(np.float64)
        ww = w[b].astype(np.float64)
        t = f32(p[:, i3[b, :, 1]] * ww[:, 1]).astype(np.float64)
        t = f32(p[:, i3[b, :, 0]] * ww[:, 0] + t).astype(np.float64)
        ref[b] = f32(p[:, i3[b, :, 2]] * ww[:, 2] + t)
    np.testing.assert_array_equal(o, ref)


def test_three_nn(oracle):
    rng = np.random.default_rng(11)
    u = scene(rng, 90)
    k = scene(rng, 50, dup=10)
    d2, idx = oracle.three_nn(u[None], k[None])
    for j in range(90):
        d = sumsq3(u[j, 0] - k[:, 0], u[j, 1] - k[:, 1], u[j, 2] - k[:, 2])
        order = np.lexsort((np.arange(50), d))[:3]      # stable: lowest index wins ties
        np.testing.assert_array_equal(idx[0, j], order)
        np.testing.assert_array_equal(d2[0, j], d[order])


def test_three_nn_fewer_than_three_known(oracle):
    d2, idx = oracle.three_nn(f32([[0, 0, 0]])[None], f32([[1, 0, 0]])[None])
    assert idx[0, 0].tolist() == [0, 0, 0] and d2[0, 0, 0] == 1.0 and np.isinf(d2[0, 0, 1:]).all()


def test_grads_are_scatter_adds(oracle):
    rng = np.random.default_rng(13)
    B, C, N, M, ns = 1, 3, 10, 6, 4
    idx = rng.integers(0, N, (B, M, ns)).astype(np.int32)
    go = f32(rng.normal(size=(B, C, M, ns)))
    g = oracle.group_points_grad(go, idx, N)
    ref = np.zeros((B, C, N), dtype=np.float64)
    for j in range(M):
        for k in range(ns):
            ref[0, :, idx[0, j, k]] += go[0, :, j, k]
    np.testing.assert_allclose(g, ref, rtol=1e-5, atol=1e-6)
    gi = rng.integers(0, N, (B, M)).astype(np.int32)
    go2 = f32(rng.normal(size=(B, C, M)))
    g2 = oracle.gather_points_grad(go2, gi, N)
    ref2 = np.zeros((B, C, N))
    for j in range(M):
        ref2[0, :, gi[0, j]] += go2[0, :, j]
    np.testing.assert_allclose(g2, ref2, rtol=1e-5, atol=1e-6)
    w = f32(rng.random((B, M, 3)))
    i3 = rng.integers(0, N, (B, M, 3)).astype(np.int32)
    g3 = oracle.three_interpolate_grad(go2, i3, w, N)
    ref3 = np.zeros((B, C, N))
    for j in range(M):
        for q in range(3):
            ref3[0, :, i3[0, j, q]] += go2[0, :, j] * w[0, j, q]
    np.testing.assert_allclose(g3, ref3, rtol=1e-5, atol=1e-6)


def test_opt_n_threads(oracle):
    # values verified against the reference formula (cuda_utils.h:13-19) in the survey
    assert [oracle.opt_n_threads(n) for n in (1, 2, 3, 511, 512, 1024, 2048, 80000)] == \
        [1, 2, 2, 256, 512, 512, 512, 512]
