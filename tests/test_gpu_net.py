"""GPU (-m gpu): the host mirror modules running on the HIP ops against outputs
of the REFERENCE's own Python modules (captured on CPU on top of the oracle
`_ext`, tests/golden/F_NET.npz).  Index outputs must be bit-equal; features go
through rocBLAS/MIOpen convolutions so they are compared to fp32 tolerance."""
import os

import numpy as np
import pytest
import torch

from rfdnet_amd import synthetic
from rfdnet_amd.iscnet.config import Config

pytestmark = pytest.mark.gpu


def close_frac(a, b, rtol=2e-3, atol=2e-4):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float((np.abs(a - b) <= atol + rtol * np.abs(b)).mean())


def sub(t, step):
    return t.detach().cpu().numpy().reshape(-1)[::step]


@pytest.fixture(scope="module")
def state(hip, golden_dir):
    fx = np.load(os.path.join(golden_dir, "F_NET.npz"))
    seed, n_raw, n_pts = (int(v) for v in fx["pc_seed"])
    pc = torch.from_numpy(synthetic.synthetic_scene(seed=seed, n_raw=n_raw, n_points=n_pts)[None]).cuda()
    return fx, pc, Config()


def test_backbone_vote_proposal_skip_match_reference(state):
    from rfdnet_amd.iscnet.pointnet2backbone import Pointnet2Backbone
    from rfdnet_amd.iscnet.proposal_module import ProposalModule
    from rfdnet_amd.iscnet.skip_propagation import SkipPropagation
    from rfdnet_amd.iscnet.vote_module import VotingModule
    fx, pc, cfg = state
    with torch.no_grad():
        bb = Pointnet2Backbone(cfg); synthetic.load_seeded(bb, 101); bb = bb.cuda().eval()
        ep = bb(pc, {})
        for k in ('sa1_inds', 'sa2_inds', 'fp2_inds'):
            np.testing.assert_array_equal(ep[k].cpu().numpy(), fx['bb_' + k])
        for k in ('sa1_xyz', 'sa2_xyz', 'sa3_xyz', 'sa4_xyz'):
            np.testing.assert_array_equal(ep[k].cpu().numpy(), fx['bb_' + k])
        for k, st in (('sa1_features', 37), ('sa2_features', 37), ('sa3_features', 17),
                      ('sa4_features', 7), ('fp2_features', 37)):
            assert close_frac(sub(ep[k], st), fx['bb_' + k]) > 0.999, k

        vote = VotingModule(cfg); synthetic.load_seeded(vote, 102); vote = vote.cuda().eval()
        vxyz, vfeat = vote(ep['fp2_xyz'], ep['fp2_features'])
        vfeat = vfeat.div(torch.norm(vfeat, p=2, dim=1).unsqueeze(1))
        assert close_frac(vxyz.cpu().numpy(), fx['vote_xyz']) > 0.999
        assert close_frac(sub(vfeat, 37), fx['vote_features']) > 0.999

        prop = ProposalModule(cfg); synthetic.load_seeded(prop, 103); prop = prop.cuda().eval()
        ep['seed_xyz'] = ep['fp2_xyz']
        ep, pf = prop(vxyz, vfeat, ep, True)
        np.testing.assert_array_equal(ep['aggregated_vote_inds'].cpu().numpy(),
                                      fx['prop_aggregated_vote_inds'])
        for k in ('aggregated_vote_xyz', 'center', 'objectness_scores', 'heading_scores',
                  'heading_residuals_normalized', 'size_scores', 'size_residuals_normalized',
                  'sem_cls_scores'):
            # a vote within 1 ulp of a ball boundary may change one neighbourhood
            assert close_frac(ep[k].cpu().numpy(), fx['prop_' + k]) > 0.99, k
        assert close_frac(sub(pf, 7), fx['prop_features']) > 0.99

        skip = SkipPropagation(cfg); synthetic.load_seeded(skip, 104); skip = skip.cuda().eval()
        ids = torch.from_numpy(fx['skip_ids']).cuda()
        centers = torch.gather(ep['center'], 1, ids.unsqueeze(-1).expand(-1, -1, 3)).contiguous()
        feats = torch.gather(pf, 2, ids.unsqueeze(1).expand(-1, 128, -1))
        ang = torch.from_numpy(fx['skip_angles']).cuda()
        codes = skip.generate(centers, ang, feats, pc)
        assert codes.shape == (1, 512, 6)
        assert close_frac(codes.cpu().numpy(), fx['skip_codes'], rtol=5e-3, atol=5e-4) > 0.98


def test_sa_module_fused_path_equals_operator_composition(hip):
    """QueryAndGroup's single fused kernel == ball_query -> group -> subtract ->
    divide -> cat built from the nine reference-named ops, bit for bit"""
    from rfdnet_amd.pointnet2_ops import pointnet2_utils as pu
    rng = np.random.default_rng(3)
    xyz = torch.from_numpy(rng.uniform(-1, 1, (2, 3000, 3)).astype(np.float32)).cuda()
    feats = torch.from_numpy(rng.normal(size=(2, 7, 3000)).astype(np.float32)).cuda()
    inds = pu.furthest_point_sample(xyz, 128)
    new_xyz = pu.gather_operation(xyz.transpose(1, 2).contiguous(), inds).transpose(1, 2).contiguous()
    qg = pu.QueryAndGroup(0.4, 16, use_xyz=True, ret_grouped_xyz=True, normalize_xyz=True)
    with torch.no_grad():
        fused, gx = qg(xyz, new_xyz, feats)
    idx = pu.ball_query(0.4, 16, xyz, new_xyz)
    g = pu.grouping_operation(xyz.transpose(1, 2).contiguous(), idx)
    g = g - new_xyz.transpose(1, 2).unsqueeze(-1)
    g = g / 0.4
    ref = torch.cat([g, pu.grouping_operation(feats, idx)], dim=1)
    assert torch.equal(fused, ref) and torch.equal(gx, g)
    # and the autograd path (training) is wired: gradients reach the features
    feats.requires_grad_(True)
    out = qg(xyz, new_xyz, feats)[0]
    out.sum().backward()
    assert feats.grad is not None and float(feats.grad.abs().sum()) > 0
