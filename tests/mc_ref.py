"""Independent numpy restatement of table-driven marching cubes (test helper).
Produces a triangle soup from the case table of tools/gen_mc_tables.py (no vertex sharing, no
ordering: used for geometric / topological properties; the ORDERED indexed mesh of the library
is oracle.marching_cubes, pinned by tests/test_mcubes_golden.py)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import gen_mc_tables as G  # noqa: E402

_TABLE = None


def table():
    global _TABLE
    if _TABLE is None:
        _TABLE = G.build()
    return _TABLE


def marching_cubes_soup(grid, iso, pad=-1e6):
    tb = table()
    g = np.pad(grid.astype(np.float64), 1, constant_values=pad)
    D = g.shape[0]
    tris = []
    C = G.CORNERS.astype(int)
    for i in range(D - 1):
        for j in range(D - 1):
            for k in range(D - 1):
                v = [g[i + c[0], j + c[1], k + c[2]] for c in C]
                ci = sum(1 << c for c in range(8) if v[c] <= iso)
                for t in tb[ci]:
                    P = []
                    for e in t:
                        a, b = G.EDGES[e]
                        lo, hi = (a, b) if tuple(C[a]) <= tuple(C[b]) else (b, a)
                        f1, f2 = v[lo], v[hi]
                        mu = 0.5 if f1 == f2 else (iso - f1) / (f2 - f1)
                        P.append(np.array([i, j, k]) + C[lo] + mu * (C[hi] - C[lo]))
                    tris.append(P)
    return np.array(tris).reshape(-1, 3, 3)


def canon(tri_xyz):
    """orientation-preserving canonical form (rotate smallest vertex first, sort)"""
    t = np.round(tri_xyz, 9)
    out = []
    for tri in t:
        keys = [tuple(p) for p in tri]
        s = keys.index(min(keys))
        out.append(keys[s] + keys[(s + 1) % 3] + keys[(s + 2) % 3])
    return sorted(out)


def index_soup(soup):
    """weld a soup into (vertices, faces)"""
    flat = np.round(soup.reshape(-1, 3), 9)
    v, inv = np.unique(flat, axis=0, return_inverse=True)
    return v, inv.reshape(-1, 3)


def assert_closed_oriented_manifold(f):
    e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
    de = set(map(tuple, e.tolist()))
    assert len(de) == e.shape[0], "a directed edge is used twice (%d of %d)" % (e.shape[0] - len(de), e.shape[0])
    assert all((b, a) in de for a, b in de), "open boundary"
