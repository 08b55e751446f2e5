"""GPU (-m gpu): device-side parse_predictions / NMS / proposal selection vs the
REFERENCE's own parse_predictions (run on CPU for tests/golden/F_NMS.npz)."""
import os

import numpy as np
import pytest
import torch

from rfdnet_amd import synthetic
from rfdnet_amd.iscnet.config import ScannetConfig

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup(hip, golden_dir):
    fn = np.load(os.path.join(golden_dir, "F_NET.npz"))
    fx = np.load(os.path.join(golden_dir, "F_NMS.npz"))
    seed, n_raw, n_pts = (int(v) for v in fn["pc_seed"])
    pc = torch.from_numpy(synthetic.synthetic_scene(seed=seed, n_raw=n_raw, n_points=n_pts)[None]).cuda()
    ep = {k[5:]: torch.from_numpy(fn[k]).cuda() for k in fn.files if k.startswith("prop_") and
          k not in ("prop_names", "prop_shapes", "prop_features", "prop_aggregated_vote_inds")}
    ep['objectness_scores'] = torch.from_numpy(fx['objectness_scores']).cuda()
    ep['size_residuals_normalized'] = torch.from_numpy(fx['size_residuals_normalized']).cuda()
    return fx, ep, pc, ScannetConfig(fx['mean_size_arr'])


@pytest.mark.parametrize("tag,cfg", [("default", {}), ("nocls", {'cls_nms': False}),
                                     ("old", {'use_old_type_nms': True}),
                                     ("keepempty", {'remove_empty_box': False})])
def test_pred_mask_matches_reference(setup, tag, cfg):
    from rfdnet_amd.iscnet import predictions
    fx, ep, pc, dc = setup
    eval_dict, parsed = predictions.parse_predictions(ep, pc, dc, cfg)
    np.testing.assert_array_equal(eval_dict['pred_mask'].cpu().numpy(), fx[tag + '_pred_mask'])
    if tag == "default":
        np.testing.assert_allclose(parsed['pred_corners_3d_upright_camera'].cpu().numpy(), fx['corners'],
                                   rtol=0, atol=1e-6)
        np.testing.assert_allclose(parsed['obj_prob'].cpu().numpy(), fx['obj_prob'], rtol=1e-6, atol=1e-7)
        ids = predictions.get_proposal_id(ep, eval_dict['pred_mask'], 0.5)
        np.testing.assert_array_equal(ids.cpu().numpy()[0, :, 0], fx['proposal_ids'])


def test_points_in_boxes_matches_delaunay_hull_test(hip):
    """the reference's in_hull (scipy Delaunay, libs.py:128-132) == oriented-box test"""
    from scipy.spatial import Delaunay
    from rfdnet_amd.iscnet import predictions
    rng = np.random.default_rng(0)
    pts = rng.uniform(-3, 3, (1, 20000, 4)).astype(np.float32)
    K = 12
    center = torch.from_numpy(rng.uniform(-2, 2, (1, K, 3))).cuda()
    size = torch.from_numpy(rng.uniform(0.3, 2.5, (1, K, 3))).cuda()
    angle = torch.from_numpy(rng.uniform(-np.pi, np.pi, (1, K))).cuda()
    boxes = torch.cat([center, size, angle.unsqueeze(-1)], -1).contiguous()
    counts = torch.empty(1, K, dtype=torch.int32, device="cuda")
    p = torch.from_numpy(pts).cuda()
    hip.check(hip.lib().rfd_points_in_boxes(1, K, 20000, 4, p.data_ptr(), boxes.data_ptr(), counts.data_ptr(),
                                            hip.current_stream()), "pib")
    corners = predictions.box_corners_upright_camera(center, size, angle).cpu().numpy()[0]
    for k in range(K):
        c = corners[k].copy()
        depth = np.stack([c[:, 0], c[:, 2], -c[:, 1]], 1)              # flip_axis_to_depth
        inside = Delaunay(depth).find_simplex(pts[0, :, :3].astype(np.float64)) >= 0
        assert abs(int(counts[0, k]) - int(inside.sum())) <= 1          # points exactly on a face


def test_generate_with_reference_selection_end_to_end(hip, setup, tmp_path):
    """`selection='nms'` (demo.py:223-256, the reference's default) through ISCNet.generate on the F_NET scene with the
    REAL class mean sizes (F_NMS.npz carries datasets/scannet/scannet_means.npz): the head outputs the fixture
    overrode are overridden the same way, and the selected proposal ids must be the ones the reference's
    parse_predictions -> get_proposal_id chose; the written obbs are those proposals' decoded boxes."""
    from rfdnet_amd import io
    from rfdnet_amd.iscnet import predictions
    from rfdnet_amd.iscnet.config import Config
    from rfdnet_amd.iscnet.network import ISCNet
    fx, ep_ref, pc, dc = setup
    cfg = Config({'generation': {'resolution_0': 8, 'upsampling_steps': 1}}, mean_size_arr=fx['mean_size_arr'])
    assert not cfg.dataset_config.placeholder_sizes
    net = ISCNet(cfg)
    for name, seed in (('backbone', 101), ('voting', 102), ('detection', 103), ('skip_propagation', 104),
                       ('completion', 105)):                          # the seeds F_NET.npz was made with
        synthetic.load_seeded(getattr(net, name), seed)
    net = net.cuda().eval()
    detect = net.detect

    def detect_like_the_fixture(point_clouds):                        # tests/golden/make_fixtures.py make_nms
        ep, pf = detect(point_clouds)
        ep['objectness_scores'] = torch.from_numpy(fx['objectness_scores']).cuda()
        ep['size_residuals_normalized'] = ep['size_residuals_normalized'] * 0.2
        return ep, pf
    net.detect = detect_like_the_fixture
    end_points, ids, meshes = net.generate({'point_clouds': pc}, selection='nms')
    np.testing.assert_array_equal(end_points['pred_mask'].cpu().numpy(), fx['default_pred_mask'])
    np.testing.assert_array_equal(ids.cpu().numpy()[0, :, 0], fx['proposal_ids'])
    assert len(meshes) == len(fx['proposal_ids'])
    box = end_points['parsed_predictions']['box_params']
    corners = predictions.box_corners_upright_camera(box[..., 0:3], box[..., 3:6], box[..., 6]).cpu().numpy()
    sel = fx['proposal_ids']
    np.testing.assert_allclose(corners[0, sel], fx['corners'][0, sel], rtol=0, atol=1e-4)
    keep = np.zeros(256, dtype=bool)
    keep[sel] = True
    io.save_visualization(str(tmp_path), pc.cpu().numpy(), ids[0].cpu().numpy(), meshes, box[0].cpu().numpy(), keep)
    written = [f for f in os.listdir(str(tmp_path)) if f.endswith('.npz')]
    assert written, os.listdir(str(tmp_path))
    d = np.load(os.path.join(str(tmp_path), written[0]))
    np.testing.assert_array_equal(d['proposal_map'][:, 0], sel)
    np.testing.assert_allclose(d['obbs'], box[0].cpu().numpy()[sel], rtol=0, atol=0)


def test_nms_selection_without_mean_sizes_fails_like_the_reference(hip):
    """scannet_config.py:21 raises on the missing scannet_means.npz; `selection='nms'` on placeholder sizes does too,
    unless the caller opts in (synthetic runs)."""
    from rfdnet_amd.iscnet.config import Config
    from rfdnet_amd.iscnet.network import ISCNet
    cfg = Config({'generation': {'resolution_0': 8, 'upsampling_steps': 1}})
    if not cfg.dataset_config.placeholder_sizes:
        pytest.skip("class mean sizes available in this environment")
    net = ISCNet(cfg)
    synthetic.load_seeded(net, 10)
    net = net.cuda().eval()
    pc = torch.from_numpy(synthetic.synthetic_scene(seed=3, n_raw=9000, n_points=8192)[None]).cuda()
    with pytest.raises(FileNotFoundError):
        net.generate({'point_clouds': pc}, selection='nms')
    cfg.eval_overrides['allow_placeholder_sizes'] = True
    with pytest.warns(RuntimeWarning):
        end_points, ids, meshes = net.generate({'point_clouds': pc}, selection='nms')
    assert end_points['pred_mask'].shape == (1, 256) and len(meshes) == ids.shape[1]
