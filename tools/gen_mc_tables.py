"""Generate the marching-cubes case table (rfdnet_amd/csrc/mc_tables.h).

The reference calls the third-party PyMCubes 0.1.2 (requirements.txt:35, call
site generator.py:160-161), which is not vendored.  Two independent sources
fix the table here:

1. derive(): for each of the 256 sign configurations the iso-polygons are
   traced around the cube face by face.  Decisions on a face depend only on that
   face's four corner signs, so two cells sharing a face always agree there =>
   the extracted surface is watertight by construction.  This fixes WHICH
   polygons a case has (incl. every ambiguous face) and their orientation.
2. CLASSIC: the published public-domain 256-row triangle list of "Polygonising a
   scalar field" (P. Bourke 1994, table by C. Bloyd) in its standard corner / edge
   numbering.  It fixes HOW each polygon is cut into triangles, the order of the
   triangles of a cell and the rotation of each index triple -- i.e. everything
   the output FACE ARRAY depends on.  That this is the table PyMCubes walks is
   not taken on faith: the meshes the reference ships under demo/outputs/ were
   written by it, 189 of the 256 rows can be read back from them
   (tests/golden/F_MC.npz) and are identical to these rows, row for row
   (tests/test_mcubes_golden.py).  The other 67 rows are configurations that do
   not occur in those meshes.

build() returns CLASSIC after checking that every row is a triangulation of
exactly the polygons derive() traces (same crossed edges, same boundary loops,
same orientation) -- all 256 agree.

Conventions (Bourke numbering): corner c at (x,y,z) below; bit c of the case
index is set when value(c) < isovalue ("outside"); triangle normals (right-hand
rule) point towards the outside (lower values).
"""
import itertools
import os

import numpy as np

CORNERS = np.array([(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0),
                    (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)], dtype=float)
EDGES = [(0, 1), (1, 2), (2, 3), (3, 0), (4, 5), (5, 6), (6, 7), (7, 4),
         (0, 4), (1, 5), (2, 6), (3, 7)]
EDGE_ID = {}
for i, (a, b) in enumerate(EDGES):
    EDGE_ID[(a, b)] = i
    EDGE_ID[(b, a)] = i
# faces: cyclic corner lists + outward normals
FACES = [((0, 1, 2, 3), (0, 0, -1)), ((4, 5, 6, 7), (0, 0, 1)),
         ((0, 1, 5, 4), (0, -1, 0)), ((3, 2, 6, 7), (0, 1, 0)),
         ((0, 3, 7, 4), (-1, 0, 0)), ((1, 2, 6, 5), (1, 0, 0))]
EDGE_MID = np.array([(CORNERS[a] + CORNERS[b]) / 2 for a, b in EDGES])


def face_segments(case, corners, normal):
    """directed segments (edge_from, edge_to) on one face"""
    s = [(case >> c) & 1 for c in corners]
    n = np.array(normal, dtype=float)
    edges = [EDGE_ID[(corners[i], corners[(i + 1) % 4])] for i in range(4)]
    crossing = [i for i in range(4) if s[i] != s[(i + 1) % 4]]
    segs = []

    def orient(ea, eb, c_set):
        A, B = EDGE_MID[ea], EDGE_MID[eb]
        side = np.dot(np.cross(B - A, CORNERS[c_set] - A), n)
        assert abs(side) > 1e-9
        return (ea, eb) if side > 0 else (eb, ea)

    if len(crossing) == 2:
        ea, eb = edges[crossing[0]], edges[crossing[1]]
        c_set = next(c for c, f in zip(corners, s) if f)
        segs.append(orient(ea, eb, c_set))
    elif len(crossing) == 4:
        # ambiguous face: cut off each SET corner separately (a rule that only
        # looks at this face, hence consistent across the two cells sharing it)
        for i in range(4):
            if s[i]:
                segs.append(orient(edges[(i - 1) % 4], edges[i], corners[i]))
    return segs


def edges_share_face(e1, e2):
    for corners, _ in FACES:
        fe = {EDGE_ID[(corners[i], corners[(i + 1) % 4])] for i in range(4)}
        if e1 in fe and e2 in fe:
            return True
    return False


def polygon_triangulations(idx):
    """all triangulations of the convex polygon with vertex list idx"""
    if len(idx) < 3:
        return [[]]
    if len(idx) == 3:
        return [[tuple(idx)]]
    out = []
    a, b = idx[0], idx[-1]
    for m in range(1, len(idx) - 1):
        for left in polygon_triangulations(idx[:m + 1]):
            for right in polygon_triangulations(idx[m:]):
                out.append(left + [(a, idx[m], b)] + right)
    return out


def best_triangulation(loop):
    """Triangulate a loop of cube-edge vertices avoiding diagonals that lie in a
    cube face (both end points on edges of one face, not adjacent in the loop):
    such a diagonal can coincide with a diagonal of the neighbouring cell and
    make the mesh non-manifold there.  Ties: prefer the plain fan."""
    n = len(loop)
    best, best_cost = None, None
    for tri in polygon_triangulations(list(range(n))):
        cost = 0
        for t in tri:
            for u, v in ((t[0], t[1]), (t[1], t[2]), (t[2], t[0])):
                if (u - v) % n in (1, n - 1):
                    continue                      # polygon side, not a diagonal
                if edges_share_face(loop[u], loop[v]):
                    cost += 1
        if best_cost is None or cost < best_cost:
            best, best_cost = tri, cost
    # keep the loop orientation: every triangle (i<j<k) in cyclic order
    return [tuple(loop[i] for i in sorted(t)) for t in best], best_cost


def triangulate(case):
    nxt = {}
    for corners, normal in FACES:
        for a, b in face_segments(case, corners, normal):
            assert a not in nxt, (case, a)
            nxt[a] = b
    tris = []
    seen = set()
    for start in sorted(nxt):
        if start in seen:
            continue
        loop = [start]
        seen.add(start)
        cur = nxt[start]
        while cur != start:
            loop.append(cur)
            seen.add(cur)
            cur = nxt[cur]
        assert len(loop) >= 3
        t, cost = best_triangulation(loop)
        triangulate.max_cost = max(getattr(triangulate, "max_cost", 0), cost)
        tris.extend(t)
    return tris


def derive():
    table = [triangulate(c) for c in range(256)]
    # global orientation: single-corner case 1 must have its normal pointing at
    # corner 0 (the set / outside corner)
    t = table[1][0]
    P = EDGE_MID[list(t)]
    nrm = np.cross(P[1] - P[0], P[2] - P[0])
    if np.dot(nrm, CORNERS[0] - P.mean(0)) < 0:
        table = [[(a, c, b) for a, b, c in tris] for tris in table]
    return table


CLASSIC = """
-
0 8 3
0 1 9
1 8 3 9 8 1
1 2 10
0 8 3 1 2 10
9 2 10 0 2 9
2 8 3 2 10 8 10 9 8
3 11 2
0 11 2 8 11 0
1 9 0 2 3 11
1 11 2 1 9 11 9 8 11
3 10 1 11 10 3
0 10 1 0 8 10 8 11 10
3 9 0 3 11 9 11 10 9
9 8 10 10 8 11
4 7 8
4 3 0 7 3 4
0 1 9 8 4 7
4 1 9 4 7 1 7 3 1
1 2 10 8 4 7
3 4 7 3 0 4 1 2 10
9 2 10 9 0 2 8 4 7
2 10 9 2 9 7 2 7 3 7 9 4
8 4 7 3 11 2
11 4 7 11 2 4 2 0 4
9 0 1 8 4 7 2 3 11
4 7 11 9 4 11 9 11 2 9 2 1
3 10 1 3 11 10 7 8 4
1 11 10 1 4 11 1 0 4 7 11 4
4 7 8 9 0 11 9 11 10 11 0 3
4 7 11 4 11 9 9 11 10
9 5 4
9 5 4 0 8 3
0 5 4 1 5 0
8 5 4 8 3 5 3 1 5
1 2 10 9 5 4
3 0 8 1 2 10 4 9 5
5 2 10 5 4 2 4 0 2
2 10 5 3 2 5 3 5 4 3 4 8
9 5 4 2 3 11
0 11 2 0 8 11 4 9 5
0 5 4 0 1 5 2 3 11
2 1 5 2 5 8 2 8 11 4 8 5
10 3 11 10 1 3 9 5 4
4 9 5 0 8 1 8 10 1 8 11 10
5 4 0 5 0 11 5 11 10 11 0 3
5 4 8 5 8 10 10 8 11
9 7 8 5 7 9
9 3 0 9 5 3 5 7 3
0 7 8 0 1 7 1 5 7
1 5 3 3 5 7
9 7 8 9 5 7 10 1 2
10 1 2 9 5 0 5 3 0 5 7 3
8 0 2 8 2 5 8 5 7 10 5 2
2 10 5 2 5 3 3 5 7
7 9 5 7 8 9 3 11 2
9 5 7 9 7 2 9 2 0 2 7 11
2 3 11 0 1 8 1 7 8 1 5 7
11 2 1 11 1 7 7 1 5
9 5 8 8 5 7 10 1 3 10 3 11
5 7 0 5 0 9 7 11 0 1 0 10 11 10 0
11 10 0 11 0 3 10 5 0 8 0 7 5 7 0
11 10 5 7 11 5
10 6 5
0 8 3 5 10 6
9 0 1 5 10 6
1 8 3 1 9 8 5 10 6
1 6 5 2 6 1
1 6 5 1 2 6 3 0 8
9 6 5 9 0 6 0 2 6
5 9 8 5 8 2 5 2 6 3 2 8
2 3 11 10 6 5
11 0 8 11 2 0 10 6 5
0 1 9 2 3 11 5 10 6
5 10 6 1 9 2 9 11 2 9 8 11
6 3 11 6 5 3 5 1 3
0 8 11 0 11 5 0 5 1 5 11 6
3 11 6 0 3 6 0 6 5 0 5 9
6 5 9 6 9 11 11 9 8
5 10 6 4 7 8
4 3 0 4 7 3 6 5 10
1 9 0 5 10 6 8 4 7
10 6 5 1 9 7 1 7 3 7 9 4
6 1 2 6 5 1 4 7 8
1 2 5 5 2 6 3 0 4 3 4 7
8 4 7 9 0 5 0 6 5 0 2 6
7 3 9 7 9 4 3 2 9 5 9 6 2 6 9
3 11 2 7 8 4 10 6 5
5 10 6 4 7 2 4 2 0 2 7 11
0 1 9 4 7 8 2 3 11 5 10 6
9 2 1 9 11 2 9 4 11 7 11 4 5 10 6
8 4 7 3 11 5 3 5 1 5 11 6
5 1 11 5 11 6 1 0 11 7 11 4 0 4 11
0 5 9 0 6 5 0 3 6 11 6 3 8 4 7
6 5 9 6 9 11 4 7 9 7 11 9
10 4 9 6 4 10
4 10 6 4 9 10 0 8 3
10 0 1 10 6 0 6 4 0
8 3 1 8 1 6 8 6 4 6 1 10
1 4 9 1 2 4 2 6 4
3 0 8 1 2 9 2 4 9 2 6 4
0 2 4 4 2 6
8 3 2 8 2 4 4 2 6
10 4 9 10 6 4 11 2 3
0 8 2 2 8 11 4 9 10 4 10 6
3 11 2 0 1 6 0 6 4 6 1 10
6 4 1 6 1 10 4 8 1 2 1 11 8 11 1
9 6 4 9 3 6 9 1 3 11 6 3
8 11 1 8 1 0 11 6 1 9 1 4 6 4 1
3 11 6 3 6 0 0 6 4
6 4 8 11 6 8
7 10 6 7 8 10 8 9 10
0 7 3 0 10 7 0 9 10 6 7 10
10 6 7 1 10 7 1 7 8 1 8 0
10 6 7 10 7 1 1 7 3
1 2 6 1 6 8 1 8 9 8 6 7
2 6 9 2 9 1 6 7 9 0 9 3 7 3 9
7 8 0 7 0 6 6 0 2
7 3 2 6 7 2
2 3 11 10 6 8 10 8 9 8 6 7
2 0 7 2 7 11 0 9 7 6 7 10 9 10 7
1 8 0 1 7 8 1 10 7 6 7 10 2 3 11
11 2 1 11 1 7 10 6 1 6 7 1
8 9 6 8 6 7 9 1 6 11 6 3 1 3 6
0 9 1 11 6 7
7 8 0 7 0 6 3 11 0 11 6 0
7 11 6
7 6 11
3 0 8 11 7 6
0 1 9 11 7 6
8 1 9 8 3 1 11 7 6
10 1 2 6 11 7
1 2 10 3 0 8 6 11 7
2 9 0 2 10 9 6 11 7
6 11 7 2 10 3 10 8 3 10 9 8
7 2 3 6 2 7
7 0 8 7 6 0 6 2 0
2 7 6 2 3 7 0 1 9
1 6 2 1 8 6 1 9 8 8 7 6
10 7 6 10 1 7 1 3 7
10 7 6 1 7 10 1 8 7 1 0 8
0 3 7 0 7 10 0 10 9 6 10 7
7 6 10 7 10 8 8 10 9
6 8 4 11 8 6
3 6 11 3 0 6 0 4 6
8 6 11 8 4 6 9 0 1
9 4 6 9 6 3 9 3 1 11 3 6
6 8 4 6 11 8 2 10 1
1 2 10 3 0 11 0 6 11 0 4 6
4 11 8 4 6 11 0 2 9 2 10 9
10 9 3 10 3 2 9 4 3 11 3 6 4 6 3
8 2 3 8 4 2 4 6 2
0 4 2 4 6 2
1 9 0 2 3 4 2 4 6 4 3 8
1 9 4 1 4 2 2 4 6
8 1 3 8 6 1 8 4 6 6 10 1
10 1 0 10 0 6 6 0 4
4 6 3 4 3 8 6 10 3 0 3 9 10 9 3
10 9 4 6 10 4
4 9 5 7 6 11
0 8 3 4 9 5 11 7 6
5 0 1 5 4 0 7 6 11
11 7 6 8 3 4 3 5 4 3 1 5
9 5 4 10 1 2 7 6 11
6 11 7 1 2 10 0 8 3 4 9 5
7 6 11 5 4 10 4 2 10 4 0 2
3 4 8 3 5 4 3 2 5 10 5 2 11 7 6
7 2 3 7 6 2 5 4 9
9 5 4 0 8 6 0 6 2 6 8 7
3 6 2 3 7 6 1 5 0 5 4 0
6 2 8 6 8 7 2 1 8 4 8 5 1 5 8
9 5 4 10 1 6 1 7 6 1 3 7
1 6 10 1 7 6 1 0 7 8 7 0 9 5 4
4 0 10 4 10 5 0 3 10 6 10 7 3 7 10
7 6 10 7 10 8 5 4 10 4 8 10
6 9 5 6 11 9 11 8 9
3 6 11 0 6 3 0 5 6 0 9 5
0 11 8 0 5 11 0 1 5 5 6 11
6 11 3 6 3 5 5 3 1
1 2 10 9 5 11 9 11 8 11 5 6
0 11 3 0 6 11 0 9 6 5 6 9 1 2 10
11 8 5 11 5 6 8 0 5 10 5 2 0 2 5
6 11 3 6 3 5 2 10 3 10 5 3
5 8 9 5 2 8 5 6 2 3 8 2
9 5 6 9 6 0 0 6 2
1 5 8 1 8 0 5 6 8 3 8 2 6 2 8
1 5 6 2 1 6
1 3 6 1 6 10 3 8 6 5 6 9 8 9 6
10 1 0 10 0 6 9 5 0 5 6 0
0 3 8 5 6 10
10 5 6
11 5 10 7 5 11
11 5 10 11 7 5 8 3 0
5 11 7 5 10 11 1 9 0
10 7 5 10 11 7 9 8 1 8 3 1
11 1 2 11 7 1 7 5 1
0 8 3 1 2 7 1 7 5 7 2 11
9 7 5 9 2 7 9 0 2 2 11 7
7 5 2 7 2 11 5 9 2 3 2 8 9 8 2
2 5 10 2 3 5 3 7 5
8 2 0 8 5 2 8 7 5 10 2 5
9 0 1 5 10 3 5 3 7 3 10 2
9 8 2 9 2 1 8 7 2 10 2 5 7 5 2
1 3 5 3 7 5
0 8 7 0 7 1 1 7 5
9 0 3 9 3 5 5 3 7
9 8 7 5 9 7
5 8 4 5 10 8 10 11 8
5 0 4 5 11 0 5 10 11 11 3 0
0 1 9 8 4 10 8 10 11 10 4 5
10 11 4 10 4 5 11 3 4 9 4 1 3 1 4
2 5 1 2 8 5 2 11 8 4 5 8
0 4 11 0 11 3 4 5 11 2 11 1 5 1 11
0 2 5 0 5 9 2 11 5 4 5 8 11 8 5
9 4 5 2 11 3
2 5 10 3 5 2 3 4 5 3 8 4
5 10 2 5 2 4 4 2 0
3 10 2 3 5 10 3 8 5 4 5 8 0 1 9
5 10 2 5 2 4 1 9 2 9 4 2
8 4 5 8 5 3 3 5 1
0 4 5 1 0 5
8 4 5 8 5 3 9 0 5 0 3 5
9 4 5
4 11 7 4 9 11 9 10 11
0 8 3 4 9 7 9 11 7 9 10 11
1 10 11 1 11 4 1 4 0 7 4 11
3 1 4 3 4 8 1 10 4 7 4 11 10 11 4
4 11 7 9 11 4 9 2 11 9 1 2
9 7 4 9 11 7 9 1 11 2 11 1 0 8 3
11 7 4 11 4 2 2 4 0
11 7 4 11 4 2 8 3 4 3 2 4
2 9 10 2 7 9 2 3 7 7 4 9
9 10 7 9 7 4 10 2 7 8 7 0 2 0 7
3 7 10 3 10 2 7 4 10 1 10 0 4 0 10
1 10 2 8 7 4
4 9 1 4 1 7 7 1 3
4 9 1 4 1 7 0 8 1 8 7 1
4 0 3 7 4 3
4 8 7
9 10 8 10 11 8
3 0 9 3 9 11 11 9 10
0 1 10 0 10 8 8 10 11
3 1 10 11 3 10
1 2 11 1 11 9 9 11 8
3 0 9 3 9 11 1 2 9 2 11 9
0 2 11 8 0 11
3 2 11
2 3 8 2 8 10 10 8 9
9 10 2 0 9 2
2 3 8 2 8 10 0 1 8 1 10 8
1 10 2
1 3 8 9 1 8
0 9 1
0 3 8
-
"""


def classic():
    rows = [r.strip() for r in CLASSIC.strip().split("\n")]
    assert len(rows) == 256
    out = []
    for r in rows:
        flat = [] if r == "-" else [int(x) for x in r.split()]
        assert len(flat) % 3 == 0
        out.append([tuple(flat[i:i + 3]) for i in range(0, len(flat), 3)])
    return out


def boundary(tris):
    """directed boundary edges of a triangle list (interior edges cancel)"""
    c = {}
    for t in tris:
        for i in range(3):
            a, b = t[i], t[(i + 1) % 3]
            if c.get((b, a), 0) > 0:
                c[(b, a)] -= 1
            else:
                c[(a, b)] = c.get((a, b), 0) + 1
    return {k for k, v in c.items() if v > 0}


def build():
    """the table the kernel and the oracle use: CLASSIC, validated against derive()"""
    tb, dv = classic(), derive()
    for case in range(256):
        assert len(tb[case]) == len(dv[case]), case
        assert boundary(tb[case]) == boundary(dv[case]), case
        assert {e for t in tb[case] for e in t} == {e for t in dv[case] for e in t}, case
    return tb


def edge_owner():
    """edge -> (dx,dy,dz, axis): the lattice point that OWNS the edge's vertex, relative to the
    cell's far corner (corner 6), and the edge's axis.  An edge belongs to its HIGH end point:
    PyMCubes creates a cell's vertices only on the three edges meeting at corner 6 (in the order
    x, y, z = edges 6, 5, 10) while it walks the cells x-major, so sorting vertices by (owner
    point, axis) reproduces its vertex numbering (read back from the reference's demo meshes,
    tests/test_mcubes_golden.py)."""
    out = []
    for a, b in EDGES:
        hi = np.maximum(CORNERS[a], CORNERS[b]).astype(int) - 1
        axis = int(np.argmax(np.abs(CORNERS[a] - CORNERS[b])))
        out.append((int(hi[0]), int(hi[1]), int(hi[2]), axis))
    return out


def main():
    table = build()
    maxt = max(len(t) for t in table)
    assert maxt <= 5, maxt
    here = os.path.dirname(os.path.abspath(__file__))
    path = os.path.join(os.path.dirname(here), "rfdnet_amd", "csrc", "mc_tables.h")
    with open(path, "w") as f:
        f.write("// GENERATED by tools/gen_mc_tables.py -- do not edit.\n"
                "// Marching-cubes case table: the published classic triangle list, validated against\n"
                "// face-loop tracing and against the reference's demo meshes (see the script).\n"
                "// bit c of the case index set <=> value(corner c) <= isovalue.\n"
                "#pragma once\n\n")
        f.write("#define MC_MAX_TRIS %d\n\n" % maxt)
        f.write("static const signed char MC_NTRI[256] = {\n")
        for r in range(16):
            f.write("  " + ", ".join(str(len(table[r * 16 + c])) for c in range(16)) + ",\n")
        f.write("};\n\n")
        f.write("static const signed char MC_TRI[256][%d] = {\n" % (3 * maxt))
        for c in range(256):
            flat = [e for t in table[c] for e in t]
            flat += [-1] * (3 * maxt - len(flat))
            f.write("  {" + ", ".join("%2d" % e for e in flat) + "},\n")
        f.write("};\n\n")
        f.write("// edge -> owning lattice point (the edge's high end) relative to the cell's FAR corner,\n"
                "// and axis (0=x,1=y,2=z)\n")
        f.write("static const signed char MC_EDGE_OWNER[12][4] = {\n")
        for o in edge_owner():
            f.write("  {%d, %d, %d, %d},\n" % o)
        f.write("};\n")
    print("in-face diagonals left (max per loop):", getattr(triangulate, "max_cost", 0))
    print("wrote", path, "max triangles per cell", maxt,
          "total triangles", sum(len(t) for t in table))
    return table


if __name__ == "__main__":
    main()
