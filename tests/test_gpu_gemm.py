"""GPU (-m gpu): split-precision GEMM (csrc/gemm_f16x3.hip) vs an fp64 reference;
fp32-class accuracy is the contract (it replaces fp32 library GEMMs)."""
import numpy as np
import pytest
import torch

from rfdnet_amd import synthetic

pytestmark = pytest.mark.gpu


def ref(x, w, bias, gb, rpg, res, relu_in, relu_out):
    a = x.double()
    if relu_in:
        a = torch.relu(a)
    y = a @ w.double().t()
    if bias is not None:
        y = y + bias.double()
    if gb is not None:
        y = y + gb.double().repeat_interleave(rpg, dim=0)
    if res is not None:
        y = y + res.double()
    if relu_out:
        y = torch.relu(y)
    return y


@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (256, 384, 160), (1024, 512, 512), (4096, 1024, 1024)])
def test_gemm_matches_fp64_reference(hip, M, N, K):
    from rfdnet_amd import gemm
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    x = torch.randn(M, K, device="cuda", generator=g) * 2.0
    w = (torch.rand(N, K, device="cuda", generator=g) * 2 - 1) / np.sqrt(K)
    bias = torch.randn(N, device="cuda", generator=g)
    y = gemm.linear(x, w, bias=bias)
    r = ref(x, w, bias, None, 1, None, False, False)
    err = (y.double() - r).abs().max().item()
    fp32 = (torch.addmm(bias, x, w.t()).double() - r).abs().max().item()
    assert err < 2e-5 * max(1.0, r.abs().max().item()), (err, fp32)
    assert err < 8 * fp32 + 1e-6, (err, fp32)          # same class as an fp32 GEMM


def test_gemm_epilogue_variants_and_strided_views(hip):
    from rfdnet_amd import gemm
    g = torch.Generator(device="cuda").manual_seed(1)
    M, N, K, T = 512, 256, 128, 64
    wide = torch.randn(M, 2 * K, device="cuda", generator=g)
    x = wide[:, :K]                                    # row view of a wider matrix (lda = 2K)
    res = torch.randn(M, 2 * N, device="cuda", generator=g)[:, N:]
    w = torch.randn(N, K, device="cuda", generator=g) * 0.1
    bias = torch.randn(N, device="cuda", generator=g)
    gb = torch.randn(M // T, N, device="cuda", generator=g)
    for relu_in in (False, True):
        for relu_out in (False, True):
            y = gemm.linear(x, w, bias=bias, gbias=gb, rows_per_group=T, residual=res,
                            relu_in=relu_in, relu_out=relu_out)
            r = ref(x, w, bias, gb, T, res, relu_in, relu_out)
            assert (y.double() - r).abs().max().item() < 2e-5 * max(1.0, r.abs().max().item())
    # transposition-detecting input: asymmetric weights, identity-like activations
    eye = torch.zeros(128, 128, device="cuda")
    eye[torch.arange(128), torch.arange(128)] = 1.0
    wa = torch.arange(128 * 128, device="cuda", dtype=torch.float32).view(128, 128) / 1000.0
    torch.testing.assert_close(gemm.linear(eye, wa), wa.t().contiguous(), rtol=1e-6, atol=1e-6)


def test_resnet_pointnet_fast_path_equals_plain(hip):
    from rfdnet_amd import synthetic
    from rfdnet_amd.iscnet.layers import ResnetPointnet
    enc = ResnetPointnet(c_dim=512, dim=132, hidden_dim=512)
    synthetic.load_seeded(enc, 9)
    enc = enc.cuda().eval()
    x = torch.randn(4, 256, 132, device="cuda")
    with torch.no_grad():
        plain = enc(x)
        fast = enc.forward_factored(enc.fc_pos(x))
    assert (plain - fast).abs().max().item() < 1e-4 * max(1.0, plain.abs().max().item())


def test_fused_group_max_pool(hip):
    """pool = max(0, C) over the rows of each group, from the epilogue of the row-owner kernel."""
    from rfdnet_amd import gemm
    g = torch.Generator(device="cuda").manual_seed(7)
    M, N, K, T = 1024, 512, 256, 128
    x = torch.randn(M, K, device="cuda", generator=g)
    w = torch.randn(N, K, device="cuda", generator=g) * 0.1
    bias = torch.randn(N, device="cuda", generator=g) - 1.0          # some columns stay negative everywhere
    res = torch.randn(M, N, device="cuda", generator=g)
    pool = torch.zeros(M // T, N, device="cuda")
    y = gemm.linear(x, w, bias=bias, residual=res, relu_in=True, rows_per_group=T, pool=pool)
    want = torch.relu(y.view(M // T, T, N).max(dim=1)[0])
    assert torch.equal(pool, want)
    r = ref(x, w, bias, None, 1, res, True, False)
    assert (y.double() - r).abs().max().item() < 2e-5 * max(1.0, r.abs().max().item())
    # pool-only launch (no product written): the cross-lane epilogue gives the same bits
    pool_only = torch.zeros(M // T, N, device="cuda")
    assert gemm.linear(x, w, bias=bias, relu_in=True, relu_out=True, rows_per_group=T, pool=pool_only,
                       store=False) is None
    y2 = gemm.linear(x, w, bias=bias, relu_in=True, relu_out=True)
    assert torch.equal(pool_only, y2.view(M // T, T, N).max(dim=1)[0])
    # shapes that only the tile kernel takes cannot pool: an error, not a silent skip
    with pytest.raises(Exception):
        gemm.linear(x[:, :96].contiguous(), w[:, :96].contiguous(), rows_per_group=T,
                    pool=torch.zeros(M // T, N, device="cuda"))


def test_pos_embed_equals_composition(hip):
    """fc_pos on cat([points, box feature]) * mask == the one-pass kernel, into a column window."""
    from rfdnet_amd import pos_embed
    g = torch.Generator(device="cuda").manual_seed(11)
    G, P, d, F_, N = 6, 192, 4, 128, 1024
    x = torch.randn(G * P, d, device="cuda", generator=g)
    mask = (torch.rand(G * P, device="cuda", generator=g) > 0.4).float()
    box = torch.randn(G, F_, device="cuda", generator=g)
    W = torch.randn(N, d + F_, device="cuda", generator=g) * 0.2
    b = torch.randn(N, device="cuda", generator=g)
    full = torch.cat([x, box.repeat_interleave(P, 0)], 1) * mask[:, None]
    want = torch.nn.functional.linear(full.double(), W.double(), b.double())
    buf = torch.full((G * P, N + 512), 7.0, device="cuda")
    out = buf[:, 512:]
    group = torch.nn.functional.linear(box, W[:, d:])
    pos_embed.pos_embed(x, mask, W, b, group, P, out)
    assert (out.double() - want).abs().max().item() < 1e-5 * max(1.0, want.abs().max().item())
    assert torch.all(buf[:, :512] == 7.0)


def test_skip_propagation_encoder_in_place_buffer(hip):
    """forward_factored on the window handed out by input_buffer() == on a plain tensor == forward()."""
    from rfdnet_amd import synthetic
    from rfdnet_amd.iscnet.layers import ResnetPointnet
    enc = ResnetPointnet(c_dim=512, dim=132, hidden_dim=512)
    synthetic.load_seeded(enc, 9)
    enc = enc.cuda().eval()
    x = torch.randn(4, 256, 132, device="cuda")
    with torch.no_grad():
        plain = enc(x)
        pos = enc.fc_pos(x)
        a = enc.forward_factored(pos)
        win = enc.input_buffer(4, 256, x.device)
        win.copy_(pos.view(-1, 1024))
        b = enc.forward_factored(win.view(4, 256, 1024))
    assert torch.equal(a, b)
    assert (plain - a).abs().max().item() < 1e-4 * max(1.0, plain.abs().max().item())


def test_fused_group_max_pool_signed(hip):
    """pool_signed: the plain max over each group's rows (columns that stay negative, zeros, mixed signs)."""
    from rfdnet_amd import gemm
    g = torch.Generator(device="cuda").manual_seed(8)
    M, N, K, T = 1024, 512, 256, 128
    x = torch.randn(M, K, device="cuda", generator=g)
    w = torch.randn(N, K, device="cuda", generator=g) * 0.05
    w[7] = 0.0                                                        # a column of exact zeros
    bias = torch.randn(N, device="cuda", generator=g) * 3.0 - 2.0     # many columns negative everywhere
    bias[7] = 0.0
    pool = torch.full((M // T, N), float("-inf"), device="cuda")
    y = gemm.linear(x, w, bias=bias, rows_per_group=T, pool=pool, pool_signed=True)
    want = y.view(M // T, T, N).max(dim=1)[0]
    assert (want < 0).any() and (want > 0).any()
    assert torch.equal(pool, want + 0.0)
    pool2 = torch.full((M // T, N), float("-inf"), device="cuda")
    assert gemm.linear(x, w, bias=bias, rows_per_group=T, pool=pool2, pool_signed=True, store=False) is None
    assert torch.equal(pool2, pool)


def test_gemm_flags_activations_beyond_the_f16_range(hip):
    """|a| * 2^4 >= 65504 saturates the f16 split: the library must say so (device status bit 2)
    instead of returning a silently wrong product.  The tile kernel watches its INPUT and its output;
    the row-owner kernel (k loop pinned instruction by instruction) and pos_embed watch what they
    STORE, i.e. the next layer's input."""
    from rfdnet_amd import gemm
    w = torch.randn(128, 32, device="cuda") * 0.1
    x = torch.zeros(128, 32, device="cuda")
    gemm.linear(x, w)
    hip.device_status()                           # clean
    x[3, 5] = -5000.0                              # negative: magnitude counts, not the sign bit
    gemm.linear(x, w)
    with pytest.raises(hip.RfdHipError, match="split-precision GEMM"):
        hip.device_status()
    x[3, 5] = 4000.0                               # 4000 * 16 = 64000 < 65504: still representable
    gemm.linear(x, w)
    hip.device_status()
    # row-owner kernel: an OUTPUT of 6000 cannot be the next layer's input
    M, N, K = 256, 256, 128
    x = torch.zeros(M, K, device="cuda")
    x[7, 0] = 100.0
    w = torch.zeros(N, K, device="cuda")
    w[11, 0] = 10.0
    y = gemm.linear(x, w)
    assert abs(float(y[7, 11]) - 1000.0) < 1e-2
    hip.device_status()
    w[11, 0] = 60.0
    y = gemm.linear(x, w)
    assert abs(float(y[7, 11]) - 6000.0) < 1e-1   # this product is still right ...
    with pytest.raises(hip.RfdHipError, match="split-precision GEMM"):
        hip.device_status()                        # ... but the next split layer would saturate on it


def test_default_stream_flags_stay_with_the_default_stream_and_slots_are_recycled(hip):
    """(i) a flag raised on the NULL stream (shared word 0) must not show up in another stream's status
    read -- round 3 folded word 0 into every stream's answer, so every scene in flight lowered its scales or failed
    until somebody cleared it; (ii) a caller that creates a stream per scene gives the slot back with
    rfd_release_stream: after far more streams than the 63 slots, two fresh streams still have words of their own."""
    from rfdnet_amd import _lib, gemm
    w = torch.randn(128, 32, device="cuda") * 0.1
    bad = torch.zeros(128, 32, device="cuda")
    bad[3, 5] = 5000.0
    good = torch.zeros(128, 32, device="cuda")
    torch.cuda.synchronize()
    gemm.linear(bad, w)                                   # default stream -> word 0
    s1 = torch.cuda.Stream()
    with torch.cuda.stream(s1):
        gemm.linear(good, w)
        assert _lib.stream_status_bits() == 0             # not s1's business
    assert _lib.stream_status_bits() == 4                 # the default stream's own read reports and clears it
    assert _lib.stream_status_bits() == 0
    _lib.release_stream(s1)
    for i in range(150):                                  # 150 streams > 63 slots: each one releases its slot
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            gemm.linear(bad if i % 50 == 7 else good, w)
            if i % 50 == 7:
                with pytest.raises(hip.RfdHipError, match="split-precision GEMM"):
                    _lib.release_stream()                 # pending flags are reported by the release, not lost
            else:
                _lib.release_stream()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.cuda.stream(sa):
        gemm.linear(bad, w)
    with torch.cuda.stream(sb):
        gemm.linear(good, w)
        assert _lib.stream_status_bits() == 0             # still isolated: sb did not fall back to the shared word
    with torch.cuda.stream(sa):
        assert _lib.stream_status_bits() == 4
    _lib.release_stream(sa)
    _lib.release_stream(sb)
    hip.device_status()


def test_a_stream_keeps_its_status_slot_when_a_lower_slot_is_released(hip):
    """rfd_release_stream leaves free slots in the middle of the table; a live stream that owns a
    HIGHER slot must keep it at its next launch (round 4 claimed the first free slot before it reached the one the
    stream already owned: flags raised by kernels in flight in the old slot were never reported to that stream, and the
    stream then held two slots)."""
    from rfdnet_amd import _lib, gemm
    w = torch.randn(128, 32, device="cuda") * 0.1
    bad = torch.zeros(128, 32, device="cuda")
    bad[3, 5] = 5000.0
    good = torch.zeros(128, 32, device="cuda")
    torch.cuda.synchronize()
    low, high = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.cuda.stream(low):
        gemm.linear(good, w)                              # claims the lower slot
    with torch.cuda.stream(high):
        gemm.linear(bad, w)                               # claims the next one and raises bit 2 in it
    _lib.release_stream(low)                              # a hole below `high`'s slot
    with torch.cuda.stream(high):
        gemm.linear(good, w)                              # must NOT move `high` into the hole ...
        assert _lib.stream_status_bits() == 4             # ... its pending flag is still its own
        assert _lib.stream_status_bits() == 0
    filler = torch.cuda.Stream()
    with torch.cuda.stream(filler):
        gemm.linear(bad, w)                               # takes the hole; `high` does not see this flag
    with torch.cuda.stream(high):
        gemm.linear(good, w)
        assert _lib.stream_status_bits() == 0
    with torch.cuda.stream(filler):
        assert _lib.stream_status_bits() == 4
    for s in (high, filler):
        _lib.release_stream(s)
    hip.device_status()


def test_two_status_snapshots_on_one_stream_do_not_overwrite_each_other(hip):
    """a caller that snapshots per stage takes a second snapshot before it has read the first; each
    must keep its own flags, and read() needs no synchronisation by the caller."""
    from rfdnet_amd import _lib, gemm
    w = torch.randn(128, 32, device="cuda") * 0.1
    bad = torch.zeros(128, 32, device="cuda")
    bad[3, 5] = 5000.0
    good = torch.zeros(128, 32, device="cuda")
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        gemm.linear(bad, w)
        first = _lib.StatusSnapshot()
        gemm.linear(good, w)
        second = _lib.StatusSnapshot()
        gemm.linear(bad, w)
        third = _lib.StatusSnapshot()
        assert (first.read(), second.read(), third.read()) == (4, 0, 4)
        assert first.read() == 4                          # reading twice is fine
        assert _lib.stream_status_bits() == 0             # every snapshot reset the word behind it
    _lib.release_stream(s)
    hip.device_status()


def test_pool_only_launch_is_range_checked_too(hip):
    """store=False launches feed the pooled maximum to the next split GEMM: the range watch covers them as well
    (round-2 advisory: `if (g.C && ...)` skipped them)."""
    from rfdnet_amd import gemm
    M, N, K, T = 256, 256, 128, 64
    x = torch.zeros(M, K, device="cuda")
    x[7, 0] = 100.0
    w = torch.zeros(N, K, device="cuda")
    w[11, 0] = 60.0
    pool = torch.zeros(M // T, N, device="cuda")
    assert gemm.pool_usable(M, N, K, T)
    gemm.linear(x, w, rows_per_group=T, pool=pool, store=False)
    assert abs(float(pool[0, 11]) - 6000.0) < 1e-1
    with pytest.raises(hip.RfdHipError, match="split-precision GEMM"):
        hip.device_status()


def test_status_words_are_per_stream(hip):
    """Scenes in flight on different streams must not see or clear each other's flags (round-2 advisory on
    rfd_stream_status): a flag raised by a launch on stream A is reported to A's status read only."""
    from rfdnet_amd import _lib, gemm
    w = torch.randn(128, 32, device="cuda") * 0.1
    bad = torch.zeros(128, 32, device="cuda")
    bad[3, 5] = 5000.0
    good = torch.zeros(128, 32, device="cuda")
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(sa):
        gemm.linear(bad, w)
    with torch.cuda.stream(sb):
        gemm.linear(good, w)
        assert _lib.stream_status_bits() == 0          # B finishes first, sees nothing and clears nothing
    with torch.cuda.stream(sa):
        assert _lib.stream_status_bits() == 4          # A's flag is still there
        assert _lib.stream_status_bits() == 0          # ... and was reset by A's own read
    hip.device_status()


def test_scene_survives_a_gemm_range_flag(hip):
    """An activation beyond the f16 range at the default GEMM scale (2^4) re-runs the stage at 2^1 instead of failing
    the scene (the reference's fp32 layers cannot overflow); the codes agree with a run that used the small scale
    from the start."""
    from rfdnet_amd import gemm
    from rfdnet_amd.iscnet.config import Config
    from rfdnet_amd.iscnet.network import ISCNet
    cfg = Config({'generation': {'resolution_0': 8, 'upsampling_steps': 0}})
    net = ISCNet(cfg)
    synthetic.load_seeded(net, 10)
    net = net.cuda().eval()
    enc = net.skip_propagation.encoder
    with torch.no_grad():
        enc.fc_pos.bias[:8] = 5000.0                    # 5000 * 2^4 > 65504 > 5000 * 2^1
    pc = torch.from_numpy(synthetic.synthetic_scene(seed=3, n_raw=9000, n_points=8192)[None]).cuda()
    assert gemm.SA == 4
    try:
        with torch.no_grad():
            ep, pf = net.detect(pc)
            ids = net.select_proposals(ep, 'all', pc)[:, :8].contiguous()
            with pytest.warns(RuntimeWarning, match="f16 range"):
                grids = net.reconstruct(ep, pf, ids, pc, return_grids=True)
            assert gemm.SA == gemm.SA_FALLBACK
            # the status word is read BEFORE the completion: the decoder never saw the clipped codes, so its own
            # activation scale is untouched (round 3 ran it on them first and could lower it for good)
            from rfdnet_amd import occ_fold
            assert net.completion.decoder.ka == occ_fold.KA
            again = net.reconstruct(ep, pf, ids, pc, return_grids=True)     # no flag, no warning, same result
        assert torch.equal(grids, again) and torch.isfinite(grids).all()
        hip.device_status()
    finally:
        gemm.SA = 4


# ---------------------------------------------------------------------------------------------------------------------
# fragment-ordered split activations (round 6): rfd_rows_to_frag / rfd_frag_to_rows / rfd_gemm_f16x3_frag /
# rfd_pos_embed_frag, and the encoder on top of them


def test_frag_rows_round_trip_and_layout(hip):
    """(hi + lo) 2^-sa reproduces relu(x) to the split's 2^-21, a re-split is bit-identical, and the layout is the one
    include/rfd_occ.h documents (checked element by element against index arithmetic on the host)"""
    from rfdnet_amd import gemm
    g = torch.Generator(device="cuda").manual_seed(3)
    M, Cc, sa = 96, 160, 4
    wide = torch.randn(M, Cc + 32, device="cuda", generator=g) * 3
    x = wide[:, 32:]                                                 # strided rows
    f = gemm.rows_to_frag(x, sa=sa)
    back = gemm.frag_to_rows(f, sa=sa)
    want = torch.relu(x)
    assert (back - want).abs().max().item() <= 2.0 ** -21 * want.abs().max().item()
    assert torch.equal(gemm.rows_to_frag(back, sa=sa), f)           # relu(relu(x)) re-splits to the same bits
    # layout: [rb][kb][kstep][split][lane][j] -> row 32 rb + (lane & 31), channel 32 kb + (r&3) + 8 (r>>2) + 4 (lane>>5)
    fh = f.float().cpu().numpy()                                     # (3, 5, 2, 2, 64, 8)
    val = (fh[:, :, :, 0] + fh[:, :, :, 1]) * 2.0 ** -sa            # (rb, kb, kstep, lane, j)
    w = want.cpu().numpy()
    rb, kb, ks, lane, j = np.meshgrid(np.arange(3), np.arange(5), np.arange(2), np.arange(64), np.arange(8), indexing="ij")
    r = 8 * ks + j
    rows = 32 * rb + (lane & 31)
    chans = 32 * kb + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    np.testing.assert_allclose(val, w[rows, chans], rtol=2.0 ** -20, atol=1e-7)
    # un-rectified conversion keeps the sign
    assert (gemm.frag_to_rows(gemm.rows_to_frag(x, sa=sa, relu=False), sa=sa) - x).abs().max().item() <= 2.0 ** -20 * 16


@pytest.mark.parametrize("M,N,K,T", [(256, 256, 128, 64), (512, 512, 384, 128), (2048, 512, 1024, 1024),
                                     # tile-list shapes of the persistent kernel: 9 m tiles (no XCD-aware list), three and
                                     # four n tiles (240 / 256 workgroups), more tiles than workgroups (3 tiles per workgroup)
                                     (2304, 768, 256, 256), (4096, 1024, 128, 512), (98304, 512, 128, 1024)])
def test_gemm_frag_matches_fp64_reference(hip, M, N, K, T):
    """rfd_gemm_f16x3_frag against fp64 on the values the frag input REALLY holds: stored output, fused pool,
    pool-only launch, channel windows on both sides -- fp32-class bounds, as for the fp32-rows kernels"""
    from rfdnet_amd import gemm
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    sa = gemm.SA
    x = torch.randn(M, K, device="cuda", generator=g) * 2.0
    w = (torch.rand(N, K, device="cuda", generator=g) * 2 - 1) / np.sqrt(K)
    bias = torch.randn(N, device="cuda", generator=g)
    gb = torch.randn(M // T, N + 64, device="cuda", generator=g)[:, 32:32 + N]     # a column window: row stride N + 64
    # A = a channel window of a wider frag buffer ([junk | A]); C = a window of another one
    abuf = gemm.frag_empty(M, K + 64, "cuda")
    abuf.view(torch.int16).fill_(0x7c00)                              # f16 inf: any read outside the window poisons
    gemm.rows_to_frag(x, sa=sa, out=abuf[:, 2:])
    a = abuf[:, 2:]
    cbuf = gemm.frag_empty(M, N + 32, "cuda")
    cbuf.view(torch.int16).fill_(0x7c00)
    pool = torch.zeros(M // T, N, device="cuda")
    out = gemm.linear_frag(a, w, bias=bias, gbias=gb, rows_per_group=T, out=cbuf[:, 1:], pool=pool, sa=sa)
    hip.device_status()
    r = ref(gemm.frag_to_rows(a, sa=sa), w, bias, gb, T, None, False, True)          # relu(A W^T + b + gb), fp64
    y = gemm.frag_to_rows(out, sa=sa)
    scale = max(1.0, r.abs().max().item())
    assert (y.double() - r).abs().max().item() < 2e-5 * scale
    assert (cbuf[:, 0].view(torch.int16) == 0x7c00).all()            # the neighbouring channel block is untouched
    rp = r.view(M // T, T, N).max(1)[0]
    assert (pool.double() - rp).abs().max().item() < 2e-5 * scale
    # pool only, signed: the plain max of the un-rectified product
    pool2 = torch.full((M // T, N), float("-inf"), device="cuda")
    assert gemm.linear_frag(a, w, bias=bias, gbias=gb, rows_per_group=T, pool=pool2, pool_signed=True, store=False,
                            sa=sa) is None
    ru = ref(gemm.frag_to_rows(a, sa=sa), w, bias, gb, T, None, False, False).view(M // T, T, N).max(1)[0]
    assert (pool2.double() - ru).abs().max().item() < 2e-5 * scale
    # no bias, no group bias, no pool
    y0 = gemm.frag_to_rows(gemm.linear_frag(a, w, sa=sa), sa=sa)
    r0 = ref(gemm.frag_to_rows(a, sa=sa), w, None, None, 1, None, False, True)
    assert (y0.double() - r0).abs().max().item() < 2e-5 * scale
    hip.device_status()


def test_gemm_frag_flags_the_f16_range(hip):
    """an OUTPUT beyond 65504 / 2^sa cannot be split for the next layer: status bit 2, as the fp32-rows kernels do"""
    from rfdnet_amd import gemm
    sa = gemm.SA
    M, N, K = 256, 256, 128
    x = torch.full((M, K), 1.0, device="cuda")
    w = torch.full((N, K), 0.25, device="cuda")
    a = gemm.rows_to_frag(x, sa=sa)
    gemm.linear_frag(a, w, sa=sa)
    hip.device_status()
    big = torch.full((N,), 65504.0 / 2 ** sa, device="cuda")
    gemm.linear_frag(a, w, bias=big, sa=sa)
    with pytest.raises(hip.RfdHipError):
        hip.device_status()
    with pytest.raises(hip.RfdHipError):                             # and rows_to_frag itself
        gemm.rows_to_frag(torch.full((32, 32), 70000.0 / 2 ** sa, device="cuda"), sa=sa)
        hip.device_status()
    hip.device_status()


@pytest.mark.parametrize("d", [4, 7])
def test_pos_embed_frag_matches_pos_embed(hip, d):
    """the frag-rows fc_pos holds exactly split(relu(fp32 fc_pos) 2^sa)"""
    from rfdnet_amd import gemm, pos_embed
    g = torch.Generator(device="cuda").manual_seed(d)
    P, K, N = 1024, 6, 1024
    M = P * K
    x = torch.randn(M, d, device="cuda", generator=g)
    mask = (torch.rand(M, device="cuda", generator=g) > 0.3).float()
    W = torch.randn(N, d + 128, device="cuda", generator=g) * 0.3
    bias = torch.randn(N, device="cuda", generator=g)
    group = torch.randn(K, N, device="cuda", generator=g)
    plain = torch.empty(M, N, device="cuda")
    pos_embed.pos_embed(x, mask, W, bias, group, P, plain)
    cat = gemm.frag_empty(M, N + 512, "cuda")
    pos_embed.pos_embed_frag(x, mask, W, bias, group, P, cat[:, 16:], gemm.SA)
    hip.device_status()
    assert torch.equal(cat[:, 16:], gemm.rows_to_frag(plain, sa=gemm.SA))            # bit for bit


def test_encoder_on_frag_rows_matches_the_module(hip):
    """ResnetPointnet.forward_frag (the round-6 path of the headline) against the module's own fp32 forward() and the
    fp32-rows factored path, on the encoder's real widths at a small proposal count"""
    from rfdnet_amd import gemm, pos_embed
    from rfdnet_amd.iscnet.layers import ResnetPointnet
    torch.manual_seed(0)
    enc = ResnetPointnet(c_dim=512, dim=132, hidden_dim=512)
    synthetic.load_seeded(enc, seed=5)
    enc = enc.cuda().eval()
    B, T, d = 4, 1024, 4
    g = torch.Generator(device="cuda").manual_seed(9)
    pts = torch.randn(B * T, d, device="cuda", generator=g)
    mask = (torch.rand(B * T, device="cuda", generator=g) > 0.2).float()
    box = torch.randn(B, 128, device="cuda", generator=g)
    w = enc.fc_pos.weight
    with torch.no_grad():
        group = torch.nn.functional.linear(box, w[:, d:])
        full = torch.cat([pts.view(B, T, d), box[:, None].expand(B, T, 128)], 2) * mask.view(B, T, 1)
        want = enc(full)                                             # the reference composition, fp32
        assert enc.frag_usable(B, T)
        sa = gemm.SA
        cat, window = enc.frag_input_buffer(B, T, "cuda")
        pos_embed.pos_embed_frag(pts, mask, w, enc.fc_pos.bias, group, T, window, sa)
        got = enc.forward_frag(cat, B, T, sa)
        pos = enc.input_buffer(B, T, "cuda")
        pos_embed.pos_embed(pts, mask, w, enc.fc_pos.bias, group, T, pos)
        rows = enc.forward_factored(pos.view(B, T, -1))
    hip.device_status()
    scale = want.abs().max().item()
    assert (got - want).abs().max().item() < 1e-4 * max(1.0, scale), (got - want).abs().max().item()
    assert (got - rows).abs().max().item() < 2e-5 * max(1.0, scale), (got - rows).abs().max().item()
