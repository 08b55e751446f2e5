"""GPU (-m gpu): the fused ResnetBlockFC kernel (csrc/resblock.hip) vs an fp64
evaluation of the reference block (layers.py:39-48) and vs the fp32 module."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _block(k_total, seed):
    from rfdnet_amd.iscnet.layers import ResnetBlockFC
    torch.manual_seed(seed)
    blk = ResnetBlockFC(k_total, 256).cuda()
    with torch.no_grad():                      # fc_1 is zero-initialised in the reference: make it count
        blk.fc_1.weight.copy_(torch.randn(256, 256, device="cuda") * 0.06)
        blk.fc_1.bias.copy_(torch.randn(256, device="cuda") * 0.1)
    return blk


def _ref64(blk, x, g0, gs, k_in, T):
    a = torch.relu(x.double())
    h = a @ blk.fc_0.weight[:, :k_in].double().t() + g0.double().repeat_interleave(T, dim=0)
    return (a @ blk.shortcut.weight[:, :k_in].double().t() + torch.relu(h) @ blk.fc_1.weight.double().t()
            + gs.double().repeat_interleave(T, dim=0))


@pytest.mark.parametrize("k_in,M,T", [(256, 128, 128), (256, 4096, 512), (512, 1024, 1024), (256, 36 * 2048, 2048),
                                       (512, 33 * 128, 33 * 128)])
def test_resblock_matches_fp64(hip, k_in, M, T):
    from rfdnet_amd import resblock
    blk = _block(512, seed=k_in + M)
    g = torch.Generator(device="cuda").manual_seed(M)
    x = torch.randn(M, k_in, device="cuda", generator=g) * 1.5
    g0 = torch.randn(M // T, 256, device="cuda", generator=g)
    gs = torch.randn(M // T, 256, device="cuda", generator=g)
    y = resblock.forward(blk, x, g0, gs, T)
    r = _ref64(blk, x, g0, gs, k_in, T)
    err = (y.double() - r).abs().max().item()
    assert err < 2e-5 * max(1.0, r.abs().max().item()), err
    from rfdnet_amd import _lib
    assert _lib.device_status() == 0


def test_resblock_equals_module_on_concatenated_input(hip):
    """Same numbers as ResnetBlockFC.forward(cat([net, pooled])) -- the way
    ResnetPointnet calls it (layers.py:380-388)."""
    import torch.nn.functional as F
    from rfdnet_amd import resblock
    blk = _block(512, seed=3)
    B, T = 6, 256
    g = torch.Generator(device="cuda").manual_seed(5)
    net = torch.randn(B, T, 256, device="cuda", generator=g)
    pooled = net.max(dim=1, keepdim=True)[0]
    with torch.no_grad():
        want = blk(torch.cat([net, pooled.expand(net.size())], dim=2))
        pr = torch.relu(pooled[:, 0])
        g0 = F.linear(pr, blk.fc_0.weight[:, 256:], blk.fc_0.bias)
        gs = F.linear(pr, blk.shortcut.weight[:, 256:], blk.fc_1.bias)
        got = resblock.forward(blk, net.view(B * T, 256), g0, gs, T).view(B, T, 256)
    assert (got - want).abs().max().item() < 5e-5 * max(1.0, want.abs().max().item())


def test_resblock_rejects_untiled_shapes(hip):
    from rfdnet_amd import resblock, _lib
    blk = _block(512, seed=1)
    x = torch.randn(100, 256, device="cuda")
    assert not resblock.usable(x, 100)
    packed, kw0, kw1 = resblock._packed(blk, 256)
    z = torch.zeros(1, 256, device="cuda")
    rc = _lib.lib().rfd_resblock_f16x3(100, 256, 100, x.data_ptr(), packed.data_ptr(), z.data_ptr(), z.data_ptr(),
                                       x.data_ptr(), kw0, kw1, _lib.current_stream())
    assert rc != 0
