"""GPU (-m gpu): the detection stage's fused glue (round 6) -- csrc/mlp_cols.hip and rfd_three_interpolate_cat --
against the torch compositions they replace (the reference's module code: PointnetFPModule
pointnet2_modules.py:345-405, VotingModule vote_module.py:34-61, ProposalModule proposal_module.py:85-124)."""
import numpy as np
import pytest
import torch

from rfdnet_amd import synthetic

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,N,widths,relus", [(1, 1024, [512, 256, 256], [1, 1]), (2, 256, [128, 128, 128, 69], [1, 1, 0]),
                                              (1, 1024, [256, 256, 256, 259], [1, 1, 0]), (3, 8, [5, 1024], [0]),
                                              (1, 40, [300, 7, 33, 64, 2], [1, 0, 1, 1])])
def test_mlp_cols_matches_fp64(hip, B, N, widths, relus):
    from rfdnet_amd import mlp
    g = torch.Generator(device="cuda").manual_seed(sum(widths))
    x = torch.randn(B, widths[0], N, device="cuda", generator=g)
    layers = []
    for cin, cout, r in zip(widths[:-1], widths[1:], relus):
        layers.append(((torch.rand(cout, cin, device="cuda", generator=g) * 2 - 1) / np.sqrt(cin),
                       torch.randn(cout, device="cuda", generator=g) * 0.3, bool(r)))
    with torch.no_grad():
        y = mlp.mlp_cols(x, layers)
    hip.device_status()
    h = x.double()
    for W, b, r in layers:
        h = torch.einsum("oc,bcn->bon", W.double(), h) + b.double()[None, :, None]
        if r:
            h = torch.relu(h)
    assert y.shape == (B, widths[-1], N)
    err = (y.double() - h).abs().max().item()
    assert err < 2e-6 * max(1.0, h.abs().max().item()), err
    with pytest.raises(AssertionError):                       # no backward: refuses under autograd
        with torch.enable_grad():
            mlp.mlp_cols(x, layers)


def test_interpolate_cat_is_the_reference_expression(hip, oracle):
    """bit-equal to the torch expressions of PointnetFPModule.forward on top of the (oracle-checked) ops"""
    from rfdnet_amd import mlp
    from rfdnet_amd.pointnet2_ops import _ext, pointnet2_utils
    g = torch.Generator(device="cuda").manual_seed(2)
    B, n, m, c, cs = 2, 1024, 512, 256, 256
    unknown = torch.rand(B, n, 3, device="cuda", generator=g) * 4
    known = unknown[:, torch.randperm(n, device="cuda", generator=g)[:m]].contiguous() + 0.01
    known[:, :5] = unknown[:, :5]                               # exact coincidences: dist 0 -> weight ~1
    feats = torch.randn(B, c, m, device="cuda", generator=g)
    skip = torch.randn(B, cs, n, device="cuda", generator=g)
    with torch.no_grad():
        dist, idx = pointnet2_utils.three_nn(unknown, known)
        dist_recip = 1.0 / (dist + 1e-8)
        weight = dist_recip / torch.sum(dist_recip, dim=2, keepdim=True)
        want = torch.cat([pointnet2_utils.three_interpolate(feats, idx, weight), skip], dim=1)
        dist2, idx2 = _ext.three_nn(unknown, known)
        got = mlp.interpolate_cat(feats, idx2, dist2, skip)
        got_noskip = mlp.interpolate_cat(feats, idx2, dist2, None)
    hip.device_status()
    assert torch.equal(idx, idx2)
    d = (got - want).abs().max().item()
    assert d <= 1e-6 * want.abs().max().item(), d             # (bit-equal when torch's sum adds left to right)
    assert torch.equal(got[:, c:], skip) and torch.equal(got_noskip, got[:, :c])


def test_fp_vote_proposal_modules_fused_against_their_torch_paths(hip):
    """the three modules at inference (fused) against the same modules' torch composition (what runs under autograd)"""
    from rfdnet_amd.iscnet.config import Config
    from rfdnet_amd.iscnet.network import ISCNet
    net = ISCNet(Config({'data': {'num_point': 20000}, 'generation': {'resolution_0': 16, 'upsampling_steps': 0}}))
    synthetic.load_seeded(net, seed=10)
    net = net.cuda().eval()
    pc = torch.from_numpy(synthetic.synthetic_scene(seed=3, n_points=20000)).cuda()[None]
    with torch.no_grad():
        fast, _ = net.detect(pc)
    with torch.enable_grad():                                  # mlp.fits() is False under autograd: the torch composition
        slow, _ = net.detect(pc)
    hip.device_status()
    for k in ('fp2_features', 'vote_xyz', 'vote_features', 'objectness_scores', 'center', 'heading_scores',
              'size_residuals_normalized', 'sem_cls_scores'):
        a, b = fast[k].detach(), slow[k].detach()
        assert a.shape == b.shape
        d = (a - b).abs().max().item()
        assert d < 2e-5 * max(1.0, b.abs().max().item()), (k, d)
    assert torch.equal(fast['aggregated_vote_inds'], slow['aggregated_vote_inds'])


def test_fold_rows_kernel_is_bit_identical_to_the_torch_composition(hip):
    """rfd_occ_fold_rows (the decoder's per-proposal table, one launch) against occ_fold.fold_table_stacked's own
    torch expressions (the path taken under autograd): the same operations in the same order -> the same bits"""
    from rfdnet_amd.iscnet.occ_decoder import DecoderCBatchNorm
    from rfdnet_amd import occ_fold
    dec = DecoderCBatchNorm(dim=3, z_dim=32, c_dim=512, hidden_size=256)
    synthetic.load_seeded(dec, 3)
    dec = dec.cuda().eval()
    g = torch.Generator(device="cuda").manual_seed(8)
    for K in (1, 13, 256):
        c = torch.randn(K, 512, device="cuda", generator=g)
        for z in (torch.zeros(K, 32, device="cuda"), torch.randn(K, 32, device="cuda", generator=g)):
            with torch.no_grad():
                t_fast, w_fast = dec.fold(z, c)
            consts = dec._fold_cache[1]
            with torch.enable_grad():
                t_slow, w_slow = occ_fold.fold_table_stacked(consts, z, c)
            assert t_fast.shape == (K, 23, 256)
            assert torch.equal(t_fast, t_slow.detach()) and torch.equal(w_fast, w_slow)
    hip.device_status()


def test_rows3_transforms_match_bmm(hip):
    from rfdnet_amd import _lib
    g = torch.Generator(device="cuda").manual_seed(6)
    G, P = 7, 1024
    rows = torch.randn(G, P, 3, device="cuda", generator=g)
    ang = torch.rand(G, device="cuda", generator=g) * 6.28
    cs = torch.stack([torch.cos(ang), torch.sin(ang)], 1).contiguous()
    out = torch.empty_like(rows)
    _lib.check(_lib.lib().rfd_rows3_rotate_z(G, P, rows.data_ptr(), cs.data_ptr(), out.data_ptr(), _lib.current_stream()), "rot")
    rot_t = torch.zeros(G, 3, 3, device="cuda", dtype=torch.float64)
    rot_t[:, 0, 0] = cs[:, 0]; rot_t[:, 1, 0] = cs[:, 1]; rot_t[:, 0, 1] = -cs[:, 1]; rot_t[:, 1, 1] = cs[:, 0]; rot_t[:, 2, 2] = 1
    assert (out.double() - torch.bmm(rows.double(), rot_t)).abs().max().item() < 1e-6
    A = torch.randn(G, 3, 4, device="cuda", generator=g)
    out2 = torch.empty_like(rows)
    _lib.check(_lib.lib().rfd_rows3_affine(G, P, rows.data_ptr(), A.data_ptr(), out2.data_ptr(), _lib.current_stream()), "aff")
    want = torch.bmm(rows.double(), A[:, :, :3].double().transpose(1, 2)) + A[:, :, 3].double().unsqueeze(1)
    assert (out2.double() - want).abs().max().item() < 2e-6
    hip.device_status()
