"""CPU: the marching-cubes table (tools/gen_mc_tables.py -> rfdnet_amd/csrc/mc_tables.h):
what any correct table must satisfy, and that the published triangle list in use cuts
exactly the polygons the face-loop derivation traces.  The rows PyMCubes actually used are
read back from the reference's demo meshes in tests/test_mcubes_golden.py."""
import os
import re

import numpy as np

import mc_ref
from mc_ref import G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_header_is_what_the_script_generates():
    src = open(os.path.join(ROOT, "rfdnet_amd", "csrc", "mc_tables.h")).read()
    rows = re.findall(r"\{([-\d,\s]+)\},", src.split("MC_TRI[256]")[1].split("};")[0])
    assert len(rows) == 256
    tb = mc_ref.table()
    for c, row in enumerate(rows):
        flat = [int(x) for x in row.split(",")]
        want = [e for t in tb[c] for e in t]
        assert flat[:len(want)] == want and all(x == -1 for x in flat[len(want):])


def test_classic_rows_triangulate_the_derived_polygons():
    tb, dv = G.classic(), G.derive()
    assert mc_ref.table() == tb
    for case in range(256):
        assert len(tb[case]) == len(dv[case])
        assert G.boundary(tb[case]) == G.boundary(dv[case]), case      # same loops, same orientation


def test_edge_owner_is_the_high_end_point():
    for e, (dx, dy, dz, axis) in enumerate(G.edge_owner()):
        a, b = G.EDGES[e]
        hi = np.maximum(G.CORNERS[a], G.CORNERS[b])
        assert tuple(hi - 1) == (dx, dy, dz) and G.CORNERS[a][axis] != G.CORNERS[b][axis]


def test_each_case_cuts_exactly_the_sign_changing_edges():
    tb = mc_ref.table()
    for case in range(256):
        crossing = {e for e, (a, b) in enumerate(G.EDGES) if ((case >> a) & 1) != ((case >> b) & 1)}
        used = {e for t in tb[case] for e in t}
        assert used == crossing, case
        assert len(tb[case]) <= 5
    assert tb[0] == [] and tb[255] == []
    assert sorted(tb[1][0]) == [0, 3, 8]            # single corner 0: the three edges at corner 0


def test_complementary_cases_have_same_edge_sets():
    tb = mc_ref.table()
    for case in range(256):
        a = {e for t in tb[case] for e in t}
        b = {e for t in tb[255 - case] for e in t}
        assert a == b


def test_watertight_and_oriented_on_random_noise():
    """white noise hits every configuration incl. all ambiguous faces"""
    rng = np.random.default_rng(0)
    for seed in range(3):
        g = rng.normal(size=(9, 9, 9)).astype(np.float32)
        soup = mc_ref.marching_cubes_soup(g, 0.0)
        v, f = mc_ref.index_soup(soup)
        mc_ref.assert_closed_oriented_manifold(f)


def test_normals_point_outside_and_volume_is_right():
    n = 14
    idx = np.stack(np.meshgrid(*[np.arange(n)] * 3, indexing="ij"), -1).astype(np.float64)
    r = 4.2
    g = (r - np.linalg.norm(idx - (n - 1) / 2, axis=-1)).astype(np.float32)       # inside positive
    soup = mc_ref.marching_cubes_soup(g, 0.0)
    vol = np.einsum('ij,ij->i', soup[:, 0], np.cross(soup[:, 1], soup[:, 2])).sum() / 6
    assert 0.93 * 4 / 3 * np.pi * r ** 3 < vol < 1.02 * 4 / 3 * np.pi * r ** 3
