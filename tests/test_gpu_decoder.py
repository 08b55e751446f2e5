"""GPU (-m gpu): fused occupancy decoder vs the oracle / reference fixture.
Tolerance: 1e-4 absolute on logits (BASELINE.json north_star)."""
import os
from collections import OrderedDict

import numpy as np
import pytest
import torch

from rfdnet_amd import synthetic

pytestmark = pytest.mark.gpu
LOGIT_TOL = 1e-4


def seeded_decoder(seed=1234):
    from rfdnet_amd.iscnet.occ_decoder import DecoderCBatchNorm
    dec = DecoderCBatchNorm(dim=3, z_dim=32, c_dim=512, hidden_size=256)
    synthetic.load_seeded(dec, seed)
    return dec.cuda().eval()


def test_decoder_matches_reference_fixture(hip, golden_dir):
    """F-DEC: logits of the reference DecoderCBatchNorm (torch CPU fp32)."""
    fx = np.load(os.path.join(golden_dir, "F_DEC.npz"))
    dec = seeded_decoder(int(fx["seed"]))
    with torch.no_grad():
        out = dec(torch.from_numpy(fx["p"]).cuda(), torch.from_numpy(fx["z"]).cuda(),
                  torch.from_numpy(fx["c"]).cuda())
    hip.device_status()
    err = np.abs(out.cpu().numpy() - fx["logits"]).max()
    assert err < LOGIT_TOL, err
    assert err < 2e-5, err            # the parity mode is fp32-class, not just in tolerance


def test_decoder_matches_oracle_ragged(hip, oracle):
    """T not a multiple of the 128-point tile, several proposals, non-zero z"""
    dec = seeded_decoder(99)
    sd = OrderedDict((k, v.detach().cpu().numpy()) for k, v in dec.state_dict().items())
    blob = oracle.decoder_param_blob(sd)
    rng = np.random.default_rng(5)
    K, T = 5, 333
    p = ((rng.random((K, T, 3)) - 0.5) * 1.1).astype(np.float32)
    z = rng.normal(0, 1, (K, 32)).astype(np.float32)
    c = rng.normal(0, 1, (K, 512)).astype(np.float32)
    ref = oracle.decoder_cbn(blob, p, z, c)
    with torch.no_grad():
        out = dec(torch.from_numpy(p).cuda(), torch.from_numpy(z).cuda(), torch.from_numpy(c).cuda())
    hip.device_status()
    assert np.abs(out.cpu().numpy() - ref).max() < LOGIT_TOL


def test_decoder_throughput_mode_error_is_reported_not_hidden(hip, golden_dir):
    from rfdnet_amd.iscnet import occ_decoder
    fx = np.load(os.path.join(golden_dir, "F_DEC.npz"))
    dec = seeded_decoder(int(fx["seed"]))
    dec.mode = occ_decoder.MODE_F16X1
    with torch.no_grad():
        out = dec(torch.from_numpy(fx["p"]).cuda(), torch.from_numpy(fx["z"]).cuda(),
                  torch.from_numpy(fx["c"]).cuda())
    err = np.abs(out.cpu().numpy() - fx["logits"]).max()
    assert err < 5e-3, err           # f16x1 is NOT the parity mode
    sign = ((out.cpu().numpy() > 0) == (fx["logits"] > 0)).mean()
    assert sign > 0.995


def test_decoder_point_independence(hip):
    """a point's logit must not depend on its tile neighbours / position"""
    dec = seeded_decoder(7)
    rng = np.random.default_rng(1)
    p = ((rng.random((1, 512, 3)) - 0.5)).astype(np.float32)
    z = np.zeros((1, 32), np.float32)
    c = rng.normal(0, 1, (1, 512)).astype(np.float32)
    with torch.no_grad():
        a = dec(torch.from_numpy(p).cuda(), torch.from_numpy(z).cuda(), torch.from_numpy(c).cuda())
        perm = rng.permutation(512)
        b = dec(torch.from_numpy(p[:, perm]).cuda(), torch.from_numpy(z).cuda(), torch.from_numpy(c).cuda())
    np.testing.assert_array_equal(a.cpu().numpy()[0][perm], b.cpu().numpy()[0])


def test_decoder_f16_overflow_falls_back_to_a_smaller_scale(hip, oracle):
    """The reference's fp32 decoder cannot overflow (occ_decoder.py:110-123).  Here activations are scaled by 2^6
    before the f16 split: |a| >= 1023.5 raises status bit 2 -- and the module answers with ONE re-run at scale 2^3
    (|a| < 8190), not with an exception; the logits must still match the oracle.  Only an overflow at the fallback
    scale raises."""
    for kern in ("w8", "w4"):
        dec = seeded_decoder(7)
        dec.kernel = kern
        with torch.no_grad():
            dec.blocks[0].bn_0.conv_beta.bias.fill_(1500.0)      # activations up to ~4800: * 2^6 > f16 max, * 2^3 fits
        rng = np.random.default_rng(2)
        p = ((rng.random((2, 256, 3)) - 0.5) * 1.1).astype(np.float32)
        z = np.zeros((2, 32), np.float32)
        c = rng.normal(0, 1, (2, 512)).astype(np.float32)
        with torch.no_grad(), pytest.warns(RuntimeWarning, match="f16 range"):
            out = dec(torch.from_numpy(p).cuda(), torch.from_numpy(z).cuda(), torch.from_numpy(c).cuda())
        assert dec.ka == 3
        hip.device_status()                                       # nothing left flagged
        sd = OrderedDict((k, v.detach().cpu().numpy()) for k, v in dec.state_dict().items())
        ref = oracle.decoder_cbn(oracle.decoder_param_blob(sd), p, z, c)
        err = np.abs(out.cpu().numpy() - ref).max() / max(1.0, np.abs(ref).max())
        print("%s: fallback scale 2^3, max |dlogit| / max(1, |logit|) = %.2e (|logit| up to %.3g)" % (kern, err, np.abs(ref).max()))
        assert err < 1e-4
        with torch.no_grad():                                     # the scale stays lowered: no second warning, same result
            again = dec(torch.from_numpy(p).cuda(), torch.from_numpy(z).cuda(), torch.from_numpy(c).cuda())
        assert torch.equal(out, again)
        with torch.no_grad():
            dec.blocks[0].bn_0.conv_beta.bias.fill_(5000.0)      # activations up to ~16000 > 8190: beyond the fallback
                                                                  # scale too -- a real error
            with pytest.raises(hip.RfdHipError, match="f16 range"):
                dec(torch.from_numpy(p).cuda(), torch.from_numpy(z).cuda(), torch.from_numpy(c).cuda())
        hip.device_status()


def test_eight_wave_kernel_gives_the_same_logits(hip, golden_dir, oracle):
    """csrc/occ_decoder8.hip (two waves per SIMD, 16x16x32 MFMA): same arithmetic, other
    fragment order -- fixture parity, oracle parity on ragged tiles, and agreement with the
    four-wave kernel to the last few ulps."""
    fx = np.load(os.path.join(golden_dir, "F_DEC.npz"))
    d4 = seeded_decoder(int(fx["seed"]))
    d8 = seeded_decoder(int(fx["seed"]))
    d8.kernel = "w8"
    p, z, c = (torch.from_numpy(fx[k]).cuda() for k in ("p", "z", "c"))
    with torch.no_grad():
        o4, o8 = d4(p, z, c), d8(p, z, c)
    hip.device_status()
    assert np.abs(o8.cpu().numpy() - fx["logits"]).max() < 2e-5
    assert (o8 - o4).abs().max().item() < 2e-6
    rng = np.random.default_rng(8)
    K, T = 3, 517
    pp = ((rng.random((K, T, 3)) - 0.5) * 1.1).astype(np.float32)
    zz = rng.normal(0, 1, (K, 32)).astype(np.float32)
    cc = rng.normal(0, 1, (K, 512)).astype(np.float32)
    sd = OrderedDict((k, v.detach().cpu().numpy()) for k, v in d8.state_dict().items())
    ref = oracle.decoder_cbn(oracle.decoder_param_blob(sd), pp, zz, cc)
    with torch.no_grad():
        out = d8(torch.from_numpy(pp).cuda(), torch.from_numpy(zz).cuda(), torch.from_numpy(cc).cuda())
    assert np.abs(out.cpu().numpy() - ref).max() < LOGIT_TOL


def test_decoder_stress_config_full_size(hip, oracle):
    """BASELINE configs[2] at full size: 256 proposals x 262 144 uniform query points (67.1 M), codes ~ N(0,1).
    Size-independent checks: (i) a point's logit does not depend on where it sits -- permuting the
    points of every proposal permutes the logits bit for bit; (ii) 1000 sampled points against the
    CPU oracle decoder (module semantics, occ_decoder.py:110-123) within 1e-4; (iii) no device
    status bit (f16 range) is raised."""
    K, T = 256, 262144
    dec = seeded_decoder(3)
    sd = OrderedDict((k, v.detach().cpu().numpy()) for k, v in dec.state_dict().items())
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    p = (torch.rand(K, T, 3, device="cuda", generator=g) - 0.5) * 1.1
    c = torch.randn(K, 512, device="cuda", generator=g)
    z = torch.zeros(K, 32, device="cuda")
    with torch.no_grad():
        out = dec(p, z, c)
        hip.device_status()
        assert out.shape == (K, T) and torch.isfinite(out).all()
        perm = torch.randperm(T, device="cuda", generator=g)
        out_p = dec(p[:, perm].contiguous(), z, c)
        hip.device_status()
    assert torch.equal(out[:, perm], out_p)
    rng = np.random.default_rng(11)
    kk = rng.integers(0, K, 1000)
    tt = rng.integers(0, T, 1000)
    order = np.argsort(kk, kind="stable")
    kk, tt = kk[order], tt[order]
    blob = oracle.decoder_param_blob(sd)
    worst = 0.0
    p_h, c_h, o_h = p[kk, tt].cpu().numpy(), c.cpu().numpy(), out[kk, tt].cpu().numpy()
    for k in np.unique(kk):
        m = kk == k
        ref = oracle.decoder_cbn(blob, p_h[m][None], np.zeros((1, 32), np.float32), c_h[k][None])
        worst = max(worst, float(np.abs(ref[0] - o_h[m]).max()))
    print("stress config: max |dlogit| on the 1000-point oracle sample = %.2e" % worst)
    assert worst < LOGIT_TOL
