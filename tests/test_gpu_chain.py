"""GPU (-m gpu): the fused PointNet feature chains (csrc/pointseg_chain.hip) against an fp64 composition of the same
layers -- [d -> 64, ReLU] -> 64 -> 128, ReLU -> 128 -> 1024 [, ReLU] -> max over each proposal's points
(models/iscnet/modules/pointseg.py:7-42, :45-79, :82-129 with the BatchNorms folded).  fp32-class accuracy is the
contract (the layers it replaces ran on the split-precision GEMM / fp32 library GEMMs)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def reference(x, l1, l2, l3, P, relu3):
    h = x.double()
    if l1 is not None:
        h = torch.relu(h @ l1[0].double().t() + l1[1].double())
    h = torch.relu(h @ l2[0].double().t() + l2[1].double())
    h = h @ l3[0].double().t() + l3[1].double()
    if relu3:
        h = torch.relu(h)
    return h.view(-1, P, l3[0].shape[0]).max(dim=1)[0]


def layers(g, d, c3=1024):
    def lin(n, k, scale):
        return ((torch.rand(n, k, device="cuda", generator=g) * 2 - 1) * scale / np.sqrt(k),
                torch.randn(n, device="cuda", generator=g) * 0.3)
    return (lin(64, d, 2.0) if d else None), lin(128, 64, 2.0), lin(c3, 128, 2.0)


@pytest.mark.parametrize("mode,d,relu3,B,P", [(1, 4, True, 3, 1024), (1, 3, True, 2, 512), (1, 7, True, 1, 1024),
                                               (2, 64, True, 3, 1024), (0, 0, False, 5, 1024), (0, 0, True, 2, 1536)])
def test_chain_matches_fp64_composition(hip, mode, d, relu3, B, P):
    from rfdnet_amd import chain
    g = torch.Generator(device="cuda").manual_seed(10 * mode + d)
    l1, l2, l3 = layers(g, d)
    if not relu3:
        l3 = (l3[0], l3[1] - 3.0)                       # negative maxima too: the pool keeps the sign
    din = d if mode else 64
    x = torch.randn(B * P, din, device="cuda", generator=g) * 1.5
    if mode == 1:                                        # a strided view like inp.reshape(B * P, D) of a wider buffer
        wide = torch.zeros(B * P, din + 3, device="cuda")
        wide[:, :din] = x
        x = wide[:, :din]
    assert chain.usable(x, P, din)
    out = chain.chain_pool(x, l1, l2, l3, P, relu3)
    hip.device_status()
    ref = reference(x, l1, l2, l3, P, relu3)
    err = (out.double() - ref).abs().max().item() / max(1.0, ref.abs().max().item())
    print("mode %d d %d P %d: max |out - fp64| / max(1, |ref|) = %.2e (|ref| up to %.2f, min %.2f)"
          % (mode, din, P, err, ref.abs().max().item(), ref.min().item()))
    assert out.shape == (B, 1024)
    if not relu3:
        assert (ref < 0).any()
    assert err < 2e-5


@pytest.mark.parametrize("c3,mode,d,relu3,B,P", [(256, 1, 3, True, 5, 1024), (64, 1, 3, True, 2, 512),
                                                  (320, 0, 0, False, 3, 1024), (960, 2, 64, True, 2, 1024)])
def test_chain_with_a_narrower_last_layer(hip, c3, mode, d, relu3, B, P):
    """round 6: the last layer's width is a run-time multiple of 64 (256 = STN_Group's STN3d,
    pointnet2_modules.py:420-466); widths whose piece count does not divide the ring's three slots included"""
    from rfdnet_amd import chain
    g = torch.Generator(device="cuda").manual_seed(c3 + d)
    l1, l2, l3 = layers(g, d, c3)
    if not relu3:
        l3 = (l3[0], l3[1] - 3.0)
    din = d if mode else 64
    x = torch.randn(B * P, din, device="cuda", generator=g) * 1.5
    out = chain.chain_pool(x, l1, l2, l3, P, relu3)
    hip.device_status()
    ref = reference(x, l1, l2, l3, P, relu3)
    assert out.shape == (B, c3)
    assert (out.double() - ref).abs().max().item() / max(1.0, ref.abs().max().item()) < 2e-5
    with pytest.raises(AssertionError):
        chain.chain_pool(x, l1, l2, (l3[0][:c3 - 4], l3[1][:c3 - 4]), P, relu3)  # not a multiple of 64


def test_stn_group_rows_path_matches_the_module(hip):
    """STN_Group.forward_rows (STN3d's conv chain + max as ONE kernel since round 6) against the module's own
    channel-major forward() -- the reference composition (pointnet2_modules.py:468-537)"""
    from rfdnet_amd import synthetic
    from rfdnet_amd.pointnet2_ops.pointnet2_modules import STN_Group
    stn = STN_Group(radius=1., nsample=1024, use_xyz=False, normalize_xyz=True)
    synthetic.load_seeded(stn, 11)
    stn = stn.cuda().eval()
    g = torch.Generator(device="cuda").manual_seed(4)
    xyz = (torch.rand(1, 20000, 3, device="cuda", generator=g) - 0.5) * 4
    feats = torch.randn(1, 2, 20000, device="cuda", generator=g)
    centres = xyz[:, :7].contiguous() + 0.05
    heading = torch.rand(1, 7, device="cuda", generator=g) * 6.28
    with torch.no_grad():
        rows, gf = stn.forward_rows(xyz, feats, centres, heading)            # (7, 1024, 3), (1, 2, 7, 1024)
        ref, gf2 = stn(xyz, feats, centres, heading)                          # (1, 3, 7, 1024)
    hip.device_status()
    assert torch.equal(gf, gf2)
    want = ref[0].permute(1, 2, 0)                                            # (7, 1024, 3)
    err = (rows - want).abs().max().item()
    print("STN_Group rows path vs module: max |d| = %.2e (|x| up to %.2f)" % (err, want.abs().max().item()))
    assert err < 1e-4 * max(1.0, want.abs().max().item())


def test_chain_is_what_the_layerwise_path_computes(hip, monkeypatch):
    """PointSeg.forward_rows with the fused chains against the same module with RFD_NO_CHAIN=1 (one GEMM per layer)."""
    from rfdnet_amd import synthetic
    from rfdnet_amd.iscnet.pointseg import PointSeg
    seg = PointSeg(2, 4)
    synthetic.load_seeded(seg, 7)
    seg = seg.cuda().eval()
    g = torch.Generator(device="cuda").manual_seed(3)
    inp = torch.randn(6, 1024, 4, device="cuda", generator=g)
    with torch.no_grad():
        a, ta = seg.forward_rows(inp)
        monkeypatch.setenv("RFD_NO_CHAIN", "1")
        b, tb = seg.forward_rows(inp)
    hip.device_status()
    d = (a - b).abs().max().item()
    dt = (ta - tb).abs().max().item()
    print("log-probabilities: max |fused - layerwise| = %.2e; feature transform: %.2e" % (d, dt))
    assert d < 1e-4 and dt < 1e-4


def test_chain_flags_activations_beyond_the_f16_range(hip):
    from rfdnet_amd import chain
    g = torch.Generator(device="cuda").manual_seed(1)
    l1, l2, l3 = layers(g, 4)
    x = torch.randn(1024, 4, device="cuda", generator=g)
    x[5, 0] = 1.0e5                                      # first-layer output ~1e5: * 2^4 leaves the f16 range
    chain.chain_pool(x, l1, l2, l3, 1024, True)
    with pytest.raises(hip.RfdHipError, match="split-precision GEMM"):
        hip.device_status()


def test_chain_pooled_output_is_range_checked_too(hip):
    """The fused chain's POOLED output feeds a split-precision GEMM directly (_TNet.forward_rows: fc1 on the pooled
    feature): like the GEMM's own pool-only path (tests/test_gpu_gemm.py::test_pool_only_launch_is_range_checked_too)
    it must raise status bit 4 when |pooled| * 2^sa leaves the f16 range although every intermediate stayed inside
    (ADVICE round 3)."""
    from rfdnet_amd import chain
    g = torch.Generator(device="cuda").manual_seed(2)
    l1, l2, l3 = layers(g, 4)
    x = torch.randn(1024, 4, device="cuda", generator=g)
    chain.chain_pool(x, l1, l2, l3, 1024, True)
    hip.device_status()                                  # clean
    w3, b3 = l3
    b3 = b3.clone()
    b3[17] = 5000.0                                      # the last layer's bias alone pushes one pooled channel to ~5000
    out = chain.chain_pool(x, l1, l2, (w3, b3), 1024, True)
    assert float(out[0, 17]) > 4095.0                    # * 2^4 > 65504
    with pytest.raises(hip.RfdHipError, match="split-precision GEMM"):
        hip.device_status()


@pytest.mark.parametrize("B,P,n_cls", [(3, 1024, 2), (2, 384, 2), (4, 1024, 1)])
def test_head_matches_fp64_composition(hip, B, P, n_cls):
    """64 -> 512 (+ per-proposal bias, ReLU) -> 256 (ReLU) -> 128 (ReLU) -> n_cls scores (pointseg.py:131-154 folded)."""
    from rfdnet_amd import chain
    g = torch.Generator(device="cuda").manual_seed(40 + P + n_cls)

    def lin(n, k, scale=2.0):
        return ((torch.rand(n, k, device="cuda", generator=g) * 2 - 1) * scale / np.sqrt(k),
                torch.randn(n, device="cuda", generator=g) * 0.3)
    (Wa, _), lb, lc, (Wd, bd) = lin(512, 64), lin(256, 512), lin(128, 256), lin(n_cls, 128)
    gbias = torch.randn(B, 512, device="cuda", generator=g)
    x = torch.randn(B * P, 64, device="cuda", generator=g) * 1.5
    assert chain.head_usable(x, P, n_cls)
    out = chain.head_scores(x, P, Wa, gbias, lb, lc, Wd, bd)
    hip.device_status()
    h = torch.relu(x.double() @ Wa.double().t() + gbias.double().repeat_interleave(P, 0))
    h = torch.relu(h @ lb[0].double().t() + lb[1].double())
    h = torch.relu(h @ lc[0].double().t() + lc[1].double())
    ref = h @ Wd.double().t() + bd.double()
    err = (out.double() - ref).abs().max().item() / max(1.0, ref.abs().max().item())
    print("head B %d P %d k %d: max |out - fp64| / max(1, |ref|) = %.2e (|ref| up to %.2f)" % (B, P, n_cls, err, ref.abs().max().item()))
    assert out.shape == (B * P, n_cls) and err < 2e-5
