"""CPU: demo-path file formats and input preparation (demo.py:24-48, 278-327)."""
import os

import numpy as np

from rfdnet_amd import io


def test_off_roundtrip_and_height_channel(tmp_path):
    rng = np.random.default_rng(0)
    v = rng.uniform(-2, 2, (500, 3))
    p = tmp_path / "scan.off"
    with open(p, "w") as f:
        f.write("OFF\n500 2 0\n")
        for r in v:
            f.write("%.9g %.9g %.9g\n" % tuple(r))
        f.write("3 0 1 2\n3 1 2 3\n")
    vv, faces = io.read_off(str(p))
    np.testing.assert_allclose(vv, v, rtol=1e-8)
    assert faces == [[0, 1, 2], [1, 2, 3]]
    data = io.load_demo_data(str(p), num_point=800, seed=10)       # fewer vertices than points: with replacement
    pc = data['point_clouds'].numpy()
    assert pc.shape == (1, 800, 4) and pc.dtype == np.float32
    floor = np.percentile(vv[:, 2], 0.99)
    np.testing.assert_allclose(pc[0, :, 3], pc[0, :, 2] - floor, atol=1e-6)   # demo.py:38-40
    assert len(np.unique(pc[0], axis=0)) < 800
    sub = io.load_demo_data(vv, num_point=100, seed=10)['point_clouds'].numpy()
    assert len(np.unique(sub[0], axis=0)) == 100                   # without replacement when enough points


def test_off_with_colours_and_glued_header(tmp_path):
    p = tmp_path / "c.off"
    with open(p, "w") as f:
        f.write("COFF\n3 1 0\n0 0 0 255 0 0 255\n1 0 0 0 255 0 255\n0 1 0 0 0 255 255\n3 0 1 2\n")
    v, faces = io.read_off(str(p))
    assert v.shape == (3, 7) and faces == [[0, 1, 2]]


def test_mesh_ply_layout_matches_the_reference_export(tmp_path):
    v = np.random.default_rng(1).normal(size=(7, 3))
    f = np.array([[0, 1, 2], [2, 3, 4], [4, 5, 6]])
    p = tmp_path / "proposal_22_mesh.ply"
    io.write_mesh_ply(str(p), v, f)
    raw = open(p, "rb").read()
    head = raw[:raw.index(b"end_header\n")].decode()
    # same element / property lines as demo/outputs/scene0549_00/proposal_22_mesh.ply
    for line in ("format binary_little_endian 1.0", "element vertex 7", "property float x", "property float y",
                 "property float z", "element face 3", "property list uchar int vertex_indices"):
        assert line in head
    assert len(raw) == raw.index(b"end_header\n") + 11 + 7 * 12 + 3 * 13
    vv, ff = io.read_mesh_ply(str(p))
    np.testing.assert_allclose(vv, v.astype(np.float32))
    np.testing.assert_array_equal(ff, f)


def test_mesh_ply_header_equals_the_reference_held_file(tmp_path):
    """F_MC.npz keeps the raw header bytes of demo/outputs/scene0549_00/proposal_107_mesh.ply and
    its arrays: writing the same mesh must give the same file up to the free-text comment line."""
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "F_MC.npz"))
    ref_head = bytes(z["ply_header"]).decode("ascii").split("\n")
    name = str(z["ply_header_of"])[:-len("_mesh.ply")]
    v, f = z[name + "_v"], z[name + "_f"]
    p = tmp_path / "m.ply"
    io.write_mesh_ply(str(p), v, f)
    raw = open(p, "rb").read()
    end = raw.index(b"end_header\n") + 11
    head = raw[:end].decode("ascii").split("\n")
    strip = lambda h: [l for l in h if not l.startswith("comment")]
    assert strip(head) == strip(ref_head)
    assert len(raw) == end + len(v) * 12 + len(f) * 13          # float xyz; uchar count + 3 int32
    vv, ff = io.read_mesh_ply(str(p))
    assert np.array_equal(vv, v) and np.array_equal(ff, f)
    # the reference-held box dump: keys, dtypes, shapes (demo.py:319-324)
    assert z["bbox_obbs"].dtype == np.float64 and z["bbox_obbs"].shape[1] == 7
    assert z["bbox_proposal_map"].dtype == np.int64 and z["bbox_proposal_map"].shape == (z["bbox_obbs"].shape[0], 1)
    assert sorted(int(n.split("_")[1]) for n in z["names"]) == sorted(z["bbox_proposal_map"][:, 0].tolist())


def test_save_visualization_files(tmp_path):
    class M(object):
        vertices = np.zeros((3, 3))
        faces = np.array([[0, 1, 2]])
    pcs = np.zeros((1, 10, 4), np.float32)
    box = np.arange(35, dtype=np.float64).reshape(5, 7)
    keep = np.array([0, 1, 0, 0, 1], bool)
    io.save_visualization(str(tmp_path), pcs, np.array([[1], [4]]), [M(), M()], box, keep)
    assert sorted(os.listdir(tmp_path)) == ['000000_pc.ply', '000000_pred_confident_nms_bbox.npz',
                                            'proposal_1_mesh.ply', 'proposal_4_mesh.ply']
    d = np.load(tmp_path / '000000_pred_confident_nms_bbox.npz')
    assert d['obbs'].shape == (2, 7) and d['proposal_map'].shape == (2, 1) and d['proposal_map'].dtype == np.int64
