"""Reading the reference-held demo meshes (tests/golden/F_MC.npz) as marching-cubes evidence.

The meshes under /root/reference/demo/outputs/scene0549_00 are what
Generator3D.extract_mesh (generator.py:145-183) produced from dense 32^3 grids with PyMCubes
0.1.2 and what demo.py:283-287 exported.  Inverting the reference's vertex transform puts
every vertex back on an edge of the padded 34^3 lattice, and from there the sign of every
lattice point, the cube index of every cell, the rows of the library's case table that were
used, and its vertex / triangle ORDER can all be read off.  A value grid that reproduces a
mesh is rebuilt too (signs by flood fill, magnitudes by weighted least squares on
log|value| from the interpolation fractions)."""
import os

import numpy as np

N = 32                    # demo grids: dense 32^3 (ISCNet_test.yaml:62-63, upsampling_steps 0)
D = N + 2
BOX = 1.1
CORNERS = [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)]
EDGES = [(0, 1), (1, 2), (2, 3), (3, 0), (4, 5), (5, 6), (6, 7), (7, 4), (0, 4), (1, 5), (2, 6), (3, 7)]
TOL = 2e-5                # float32 PLY coordinates: ~4e-6 cells of rounding


def load():
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "F_MC.npz"))
    return z, [str(n) for n in z["names"]]


def to_lattice(v, offset=1.5):
    """invert generator.py:163-168: padded-grid index coordinates of canonical vertices"""
    return (np.asarray(v, np.float64) / BOX + 0.5) * (N - 1) + offset


def vertex_edges(u):
    """-> base (nv,3) int, axis (nv,), t (nv,): vertex i = base + t * e_axis, 0 <= t <= 1"""
    r = u - np.round(u)
    isint = np.abs(r) < TOL
    nint = isint.sum(1)
    assert (nint >= 2).all(), "a vertex is not on a lattice edge"
    base = np.round(u).astype(int)
    axis = np.where(nint == 2, np.argmin(isint, axis=1), -1)
    for i in np.where(nint == 3)[0]:
        # on a lattice point to float32 precision: a crossing of the padding shell (the interior
        # point carries the vertex) or a tiny interpolation fraction
        P = base[i]
        shell = [a for a in range(3) if P[a] in (1, D - 2)]
        if len(shell) == 1:
            a = shell[0]
            base[i, a] = 0 if P[a] == 1 else P[a]
            if P[a] == 1:
                pass                     # edge from the low shell point 0 to 1
        else:
            a = int(np.argmax(np.abs(r[i])))
            base[i, a] = int(np.floor(u[i, a]))
        axis[i] = a
    two = nint == 2
    idx = np.where(two)[0]
    base[idx, axis[idx]] = np.floor(u[idx, axis[idx]]).astype(int)
    t = u[np.arange(len(u)), axis] - base[np.arange(len(u)), axis]
    return base, axis, np.clip(t, 0.0, 1.0)


def flood_signs(base, axis):
    """inside mask (D,D,D) from the set of crossed edges: lattice neighbours have equal sign
    iff no vertex sits between them; (0,0,0) is padding = outside"""
    crossed = np.zeros((3, D, D, D), bool)
    crossed[axis, base[:, 0], base[:, 1], base[:, 2]] = True
    lab = -np.ones((D, D, D), np.int8)
    lab[0, 0, 0] = 0
    stack = [(0, 0, 0)]
    while stack:
        p = stack.pop()
        for a in range(3):
            for d in (-1, 1):
                q = list(p)
                q[a] += d
                if not 0 <= q[a] < D:
                    continue
                q = tuple(q)
                lo = p if d == 1 else q
                s = lab[p] ^ int(crossed[a][lo])
                if lab[q] < 0:
                    lab[q] = s
                    stack.append(q)
                else:
                    assert lab[q] == s, "crossed edges do not bound a region"
    return lab == 1


def cube_indices(inside):
    """(D-1)^3 cube indices, bit c set <=> corner c outside (below the iso level)"""
    out = ~inside
    ci = np.zeros((D - 1,) * 3, np.int32)
    for c, (x, y, z) in enumerate(CORNERS):
        ci |= out[x:x + D - 1, y:y + D - 1, z:z + D - 1].astype(np.int32) << c
    return ci


def edge_key(cell, e):
    a, b = EDGES[e]
    pa, pb = np.add(CORNERS[a], cell), np.add(CORNERS[b], cell)
    lo = np.minimum(pa, pb)
    return (int(lo[0]), int(lo[1]), int(lo[2]), int(np.argmax(pa != pb)))


def faces_by_cell(base, axis, faces):
    """-> list of (cell, [local-edge triples in file order]) in file order"""
    key_of = [(int(b[0]), int(b[1]), int(b[2]), int(a)) for b, a in zip(base, axis)]

    def cells_of(k):
        x, y, z, a = k
        o = [b for b in range(3) if b != a]
        cs = set()
        for d0 in (0, -1):
            for d1 in (0, -1):
                c = [x, y, z]
                c[o[0]] += d0
                c[o[1]] += d1
                if all(0 <= c[b] < D - 1 for b in range(3)):
                    cs.add(tuple(c))
        return cs

    out = []
    for tr in faces:
        cs = cells_of(key_of[tr[0]]) & cells_of(key_of[tr[1]]) & cells_of(key_of[tr[2]])
        assert len(cs) == 1, "a face does not lie in one cell"
        cell = next(iter(cs))
        loc = {edge_key(cell, e): e for e in range(12)}
        tri = tuple(loc[key_of[x]] for x in tr)
        if out and out[-1][0] == cell:
            out[-1][1].append(tri)
        else:
            out.append((cell, [tri]))
    return out


def rebuild_grid(base, axis, t, inside):
    """an (N,N,N) float32 value grid whose iso-0 surface is the mesh: sign from `inside`,
    log-magnitudes from |v_lo| / |v_hi| = t / (1 - t) on every crossed interior edge
    (weighted least squares; the fractions carry float32 noise)."""
    from scipy.sparse import coo_matrix
    from scipy.sparse.linalg import lsqr
    pid = -np.ones((D, D, D), np.int64)
    rows, cols, vals, rhs = [], [], [], []
    n_eq = 0
    n_unk = 0

    def unk(p):
        nonlocal n_unk
        if pid[p] < 0:
            pid[p] = n_unk
            n_unk += 1
        return pid[p]

    for b, a, tt in zip(base, axis, t):
        lo = tuple(int(x) for x in b)
        hi = list(lo)
        hi[a] += 1
        hi = tuple(hi)
        if min(lo) < 1 or max(hi) > D - 2:
            continue                                  # padding-shell edge: no constraint
        tt = min(max(tt, 1e-7), 1 - 1e-7)
        w = tt * (1 - tt) / 4e-6
        rows += [n_eq, n_eq]
        cols += [unk(lo), unk(hi)]
        vals += [w, -w]
        rhs.append(w * np.log(tt / (1 - tt)))
        n_eq += 1
    A = coo_matrix((vals, (rows, cols)), shape=(n_eq, n_unk)).tocsr()
    x = lsqr(A, np.array(rhs), atol=1e-14, btol=1e-14, iter_lim=20000)[0]
    x -= x.max()
    mag = np.ones((D, D, D))
    sel = pid >= 0
    mag[sel] = np.exp(x[pid[sel]])
    g = np.where(inside, mag, -mag)[1:-1, 1:-1, 1:-1]
    return np.ascontiguousarray(g, dtype=np.float32)


def analyse(v, f):
    u = to_lattice(v)
    base, axis, t = vertex_edges(u)
    inside = flood_signs(base, axis)
    return u, base, axis, t, inside
