"""Generate the marching-cubes case table (rfdnet_amd/csrc/mc_tables.h).

The reference calls the third-party PyMCubes 0.1.2 (requirements.txt:35, call
site generator.py:160-161), which is not vendored, so its 256-entry table cannot
be consulted.  This script DERIVES a table instead of recalling one: for each
of the 256 sign configurations the iso-polygons are traced around the cube face
by face.  Decisions on a face depend only on that face's four corner signs, so
two cells sharing a face always agree there => the extracted surface is
watertight by construction (checked in tests/test_mcubes_table.py).

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


def build():
    table = [triangulate(c) for c in range(256)]
    # global orientation: single-corner case 1 must have its normal pointing at
    # corner 0 (the set / outside corner)
    t = table[1][0]
    P = EDGE_MID[list(t)]
    nrm = np.cross(P[1] - P[0], P[2] - P[0])
    if np.dot(nrm, CORNERS[0] - P.mean(0)) < 0:
        table = [[(a, c, b) for a, b, c in tris] for tris in table]
    return table


def edge_owner():
    """edge -> (dx,dy,dz of the owning lattice point, axis)"""
    out = []
    for a, b in EDGES:
        lo = np.minimum(CORNERS[a], CORNERS[b]).astype(int)
        axis = int(np.argmax(np.abs(CORNERS[a] - CORNERS[b])))
        out.append((int(lo[0]), int(lo[1]), int(lo[2]), axis))
    return out


def main():
    table = build()
    maxt = max(len(t) for t in table)
    assert maxt <= 5, maxt
    here = os.path.dirname(os.path.abspath(__file__))
    path = os.path.join(os.path.dirname(here), "rfdnet_amd", "csrc", "mc_tables.h")
    with open(path, "w") as f:
        f.write("// GENERATED by tools/gen_mc_tables.py -- do not edit.\n"
                "// Marching-cubes case table derived by face-loop tracing (see the script).\n"
                "// bit c of the case index set <=> value(corner c) < isovalue.\n"
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
        f.write("// edge -> owning lattice point offset (dx,dy,dz) and axis (0=x,1=y,2=z)\n")
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
