"""fit_mesh_to_scan (network.py:182-303): box helpers on CPU; on the GPU the whole refinement
against the reference's own function run on CPU with the reference's CPU Chamfer op
(tests/golden/F_FIT.npz, make_fixtures.py fit)."""
import os

import numpy as np
import pytest
import torch

from rfdnet_amd.iscnet import fit, predictions


def test_get_3d_box_matches_the_prediction_decoder_and_inverts():
    g = torch.Generator().manual_seed(0)
    center = torch.randn(2, 5, 3, generator=g, dtype=torch.float64)
    size = torch.rand(2, 5, 3, generator=g, dtype=torch.float64) + 0.3
    angle = (torch.rand(2, 5, generator=g, dtype=torch.float64) - 0.5) * 6.0
    want = predictions.box_corners_upright_camera(center, size, angle)        # pinned by F_NMS
    got = fit.get_3d_box(size, -angle, fit.flip_axis_to_camera(center))
    assert torch.allclose(got, want, atol=1e-12)
    c, s, o = fit.box_params_from_corners(got.view(-1, 8, 3))
    assert torch.allclose(c, center.view(-1, 3), atol=1e-12) and torch.allclose(s, size.view(-1, 3), atol=1e-12)
    assert torch.allclose(torch.cos(o), torch.cos(angle.view(-1)), atol=1e-12)
    assert torch.allclose(torch.sin(o), torch.sin(angle.view(-1)), atol=1e-12)
    assert torch.allclose(fit.flip_axis_to_depth(fit.flip_axis_to_camera(center)), center)


def test_points_in_box_is_an_oriented_box_test():
    g = torch.Generator().manual_seed(1)
    center = torch.tensor([[0.3, -0.2, 0.5]], dtype=torch.float64)
    size = torch.tensor([[1.0, 0.6, 0.8]], dtype=torch.float64)
    heading = torch.tensor([0.7], dtype=torch.float64)
    corners = fit.flip_axis_to_depth(fit.get_3d_box(size, -heading, fit.flip_axis_to_camera(center)))[0]
    pts = (torch.rand(4000, 3, generator=g, dtype=torch.float64) - 0.5) * 3
    rel = pts - center
    c, s = torch.cos(heading), torch.sin(heading)
    lx = c * rel[:, 0] + s * rel[:, 1]
    ly = -s * rel[:, 0] + c * rel[:, 1]
    want = (lx.abs() <= 0.5) & (ly.abs() <= 0.3) & (rel[:, 2].abs() <= 0.4)
    got = fit.points_in_box(pts, corners)
    assert torch.equal(got, want) and 50 < int(want.sum()) < 3900


def test_normalise_mesh_points_unit_extent_in_shapenet_frame():
    v = torch.tensor(np.random.default_rng(2).uniform(-1, 2, (500, 3)))
    o = fit.normalise_mesh_points(v)
    assert torch.allclose(o.max(0)[0] - o.min(0)[0], torch.ones(3, dtype=torch.float64))
    assert torch.allclose(o.max(0)[0] + o.min(0)[0], torch.zeros(3, dtype=torch.float64), atol=1e-12)
    # axis permutation of transform_shapenet: new x = -old z, new y = -old x, new z = old y
    k = int(v[:, 2].argmax())
    assert o[k, 0] == o[:, 0].min()


@pytest.mark.gpu
def test_fit_mesh_to_scan_matches_reference_run(golden_dir, hip):
    fx = np.load(os.path.join(golden_dir, "F_FIT.npz"))
    K = int(fx["n_meshes"])

    class M(object):
        pass
    meshes = []
    for j in range(K):
        m = M()
        m.vertices = fx["verts_%d" % j]
        meshes.append(m)
    parsed = {'pred_corners_3d_upright_camera': torch.from_numpy(fx["corners_in"]).cuda(),
              'obj_prob': torch.from_numpy(fx["obj_prob"]).cuda()}
    out = fit.fit_mesh_to_scan(meshes, np.arange(K).reshape(1, K, 1), parsed,
                               {'pred_mask': torch.from_numpy(fx["pred_mask"]).cuda()},
                               torch.from_numpy(fx["scan"]).cuda(), 0.5)
    got = out['pred_corners_3d_upright_camera'].cpu().numpy()
    want = fx["corners_out"]
    moved = np.abs(want - fx["corners_in"]).reshape(K, -1).max(1)
    assert moved[:2].min() > 0.02 and moved[2] == 0          # two boxes refined, the masked one untouched
    assert np.array_equal(got[0, 2], fx["corners_in"][0, 2])
    # same optimisation, float atomics in the gradient: centimetre-level agreement on boxes ~1 m
    assert np.abs(got - want).max() < 5e-3, np.abs(got - want).reshape(K, -1).max(1)
    assert out['fit_indices'] == [(0, 0), (0, 1)]
