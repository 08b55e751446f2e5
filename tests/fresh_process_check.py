"""One FRESH process = one fresh GPU context: run the hand-scheduled kernels once each against
their references and exit non-zero on any mismatch.  Driven ~20 times by
tests/test_gpu_fresh_process.py -- a fused ResnetBlockFC kernel withdrawn in round 1 produced a wrong
tile in 1 of 14 fresh-process runs while every warm loop was clean; this is the net for that
failure mode (a register read before its hand-issued load has landed shows up only with cold
caches / first-touch page faults).

  kernels: gemm_rows8 (row-owner split-precision GEMM, with and without fused pooling),
           the tile GEMM, occ_decode (4 waves), occ_decode8 (8 waves), sa_fused, the fused PointNet chains,
           and (round 5) furthest point sampling: the multi-workgroup exchange (one polling wave per workgroup,
           tagged granules through memory, hand-written DPP reductions) and a single-workgroup level.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from rfdnet_amd import _lib, gemm, synthetic  # noqa: E402


def check_gemm(seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    worst = 0.0
    # (M, N, K): row-owner kernel shapes (M,N % 256, K % 128) and a tile-kernel shape
    for M, N, K, T in ((2048, 512, 512, 256), (1024, 1024, 1024, 1024), (1024, 256, 128, 128), (384, 128, 96, 0)):
        x = torch.randn(M, K, device="cuda", generator=g) * 2.0
        w = (torch.rand(N, K, device="cuda", generator=g) * 2 - 1) / np.sqrt(K)
        bias = torch.randn(N, device="cuda", generator=g)
        r = torch.relu(x.double()) @ w.double().t() + bias.double()
        y = gemm.linear(x, w, bias=bias, relu_in=True)
        e = (y.double() - r).abs().max().item() / max(1.0, r.abs().max().item())
        worst = max(worst, e)
        if T and gemm.pool_usable(M, N, K, T):
            pool = torch.zeros(M // T, N, device="cuda")
            gemm.linear(x, w, bias=bias, relu_in=True, pool=pool, rows_per_group=T, store=False)
            rp = torch.relu(r).view(M // T, T, N).max(dim=1)[0]
            worst = max(worst, (pool.double() - rp).abs().max().item() / max(1.0, rp.abs().max().item()))
    assert worst < 2e-5, "split-precision GEMM off by %.3g (relative)" % worst
    return worst


def check_decoders():
    from rfdnet_amd.iscnet.occ_decoder import DecoderCBatchNorm
    fx = np.load(os.path.join(ROOT, "tests", "golden", "F_DEC.npz"))
    out = {}
    for kern in ("w4", "w8"):
        dec = DecoderCBatchNorm(dim=3, z_dim=32, c_dim=512, hidden_size=256)
        synthetic.load_seeded(dec, int(fx["seed"]))
        dec = dec.cuda().eval()
        dec.kernel = kern
        with torch.no_grad():
            o = dec(torch.from_numpy(fx["p"]).cuda(), torch.from_numpy(fx["z"]).cuda(), torch.from_numpy(fx["c"]).cuda())
        err = float(np.abs(o.cpu().numpy() - fx["logits"]).max())
        assert err < 2e-5, "decoder %s off by %.3g vs the reference fixture" % (kern, err)
        out[kern] = err
        # several proposals, ragged tiles, twice: a race shows up as a wrong 16-point group in ONE of the runs
        # (the static-priority build of round 2 failed this in 80 % of the processes and nothing else did)
        from collections import OrderedDict
        from oracle import oracle
        sd = OrderedDict((k, v.detach().cpu().numpy()) for k, v in dec.state_dict().items())
        blob = oracle.decoder_param_blob(sd)
        rng = np.random.default_rng(5)
        for K, T in ((5, 333), (8, 1024)):
            p = ((rng.random((K, T, 3)) - 0.5) * 1.1).astype(np.float32)
            z = rng.normal(0, 1, (K, 32)).astype(np.float32)
            c = rng.normal(0, 1, (K, 512)).astype(np.float32)
            ref = oracle.decoder_cbn(blob, p, z, c)
            with torch.no_grad():
                a = dec(torch.from_numpy(p).cuda(), torch.from_numpy(z).cuda(), torch.from_numpy(c).cuda()).cpu().numpy()
                b = dec(torch.from_numpy(p).cuda(), torch.from_numpy(z).cuda(), torch.from_numpy(c).cuda()).cpu().numpy()
            assert np.array_equal(a, b), "decoder %s: two runs on the same input differ by %.3g" % (kern, np.abs(a - b).max())
            e = float(np.abs(a - ref).max())
            assert e < 1e-5, "decoder %s off by %.3g vs the oracle on %d x %d points" % (kern, e, K, T)
    return out


def check_decoder_stress():
    """configs[2] size from a COLD context (three of the twenty processes): 256 proposals x 262 144 points through the
    eight-wave kernel twice -- the two runs must be bit-identical (a per-wave race shows as one 16-point group
    differing), and a 1000-point sample must match the oracle."""
    from collections import OrderedDict
    from oracle import oracle
    from rfdnet_amd.iscnet.occ_decoder import DecoderCBatchNorm
    dec = DecoderCBatchNorm(dim=3, z_dim=32, c_dim=512, hidden_size=256)
    synthetic.load_seeded(dec, 3)
    dec = dec.cuda().eval()
    K, T = 256, 262144
    g = torch.Generator(device="cuda").manual_seed(11)
    pts = ((torch.rand(K * T, 3, device="cuda", generator=g) - 0.5) * 1.1).contiguous()
    c = torch.randn(K, 512, device="cuda", generator=g)
    z = torch.zeros(K, 32, device="cuda")
    with torch.no_grad():
        table, fcp = dec.fold(z, c)
        tile_prop = torch.arange(K, dtype=torch.int32, device="cuda").repeat_interleave(T // 128)
        a = dec.decode_tiles(pts, tile_prop, table, fcp)
        b = dec.decode_tiles(pts, tile_prop, table, fcp)
    assert torch.equal(a, b), "stress-size decode: two launches differ in %d points" % int((a != b).sum())
    sd = OrderedDict((k, v.detach().cpu().numpy()) for k, v in dec.state_dict().items())
    blob = oracle.decoder_param_blob(sd)
    rng = np.random.default_rng(1)
    ks = rng.integers(0, K, 1000)
    ts = rng.integers(0, T, 1000)
    p = pts.view(K, T, 3)[torch.from_numpy(ks).cuda(), torch.from_numpy(ts).cuda()].cpu().numpy()
    ref = np.concatenate([oracle.decoder_cbn(blob, p[i:i + 1, None], np.zeros((1, 32), np.float32),
                                             c[ks[i]:ks[i] + 1].cpu().numpy())[0] for i in range(0, 1000)])
    got = a.view(K, T)[torch.from_numpy(ks).cuda(), torch.from_numpy(ts).cuda()].cpu().numpy()
    e = float(np.abs(got - ref).max())
    assert e < 1e-5, "stress-size decode off by %.3g vs the oracle" % e
    return e


def check_chain(seed):
    """the fused PointNet feature chains (csrc/pointseg_chain.hip: LDS-DMA weight ring, hand-counted vmcnt waits)
    from a cold context, all three modes, twice each: bit-identical runs and fp32-class agreement with fp64"""
    from rfdnet_amd import chain
    g = torch.Generator(device="cuda").manual_seed(100 + seed)

    def lin(n, k):
        return ((torch.rand(n, k, device="cuda", generator=g) * 2 - 1) * 2.0 / np.sqrt(k),
                torch.randn(n, device="cuda", generator=g) * 0.3)
    worst = 0.0
    for d, relu3 in ((4, True), (64, True), (0, False)):
        l1 = lin(64, d) if d else None
        l2, l3 = lin(128, 64), lin(1024, 128)
        x = torch.randn(3 * 1024, d if d else 64, device="cuda", generator=g)
        a = chain.chain_pool(x, l1, l2, l3, 1024, relu3)
        b = chain.chain_pool(x, l1, l2, l3, 1024, relu3)
        assert torch.equal(a, b), "fused chain (d=%d): two runs on the same input differ" % d
        h = x.double()
        if l1 is not None:
            h = torch.relu(h @ l1[0].double().t() + l1[1].double())
        h = torch.relu(h @ l2[0].double().t() + l2[1].double()) @ l3[0].double().t() + l3[1].double()
        if relu3:
            h = torch.relu(h)
        ref = h.view(3, 1024, 1024).max(dim=1)[0]
        worst = max(worst, (a.double() - ref).abs().max().item() / max(1.0, ref.abs().max().item()))
    # ... and PointSeg's fused head (same ring machinery, 40-KiB pieces)
    (Wa, _), lb, lc, (Wd, bd) = lin(512, 64), lin(256, 512), lin(128, 256), lin(2, 128)
    gbias = torch.randn(3, 512, device="cuda", generator=g)
    x = torch.randn(3 * 1024, 64, device="cuda", generator=g)
    a = chain.head_scores(x, 1024, Wa, gbias, lb, lc, Wd, bd)
    b = chain.head_scores(x, 1024, Wa, gbias, lb, lc, Wd, bd)
    assert torch.equal(a, b), "fused head: two runs on the same input differ"
    h = torch.relu(x.double() @ Wa.double().t() + gbias.double().repeat_interleave(1024, 0))
    h = torch.relu(torch.relu(h @ lb[0].double().t() + lb[1].double()) @ lc[0].double().t() + lc[1].double())
    ref = h @ Wd.double().t() + bd.double()
    worst = max(worst, (a.double() - ref).abs().max().item() / max(1.0, ref.abs().max().item()))
    assert worst < 2e-5, "fused chain / head off by %.3g (relative)" % worst
    return worst


def check_fps(seed):
    """SA1 of the headline scene (80 000 -> 2048, 32 workgroups exchanging candidates every round) from a cold context,
    twice: bit-equal to the reference-run fixture F_NET80k (the reference's backbone on the oracle ops) and to itself;
    a single-workgroup level (2048 -> 512) against the oracle."""
    from oracle import oracle
    from rfdnet_amd.pointnet2_ops import _ext
    fx = np.load(os.path.join(ROOT, "tests", "golden", "F_NET80k.npz"))
    sc_seed, n_raw, n_pts = (int(v) for v in fx["pc_seed"])
    pc = synthetic.synthetic_scene(seed=sc_seed, n_raw=n_raw, n_points=n_pts)
    xyz = np.ascontiguousarray(pc[None, :, :3])
    x = torch.from_numpy(xyz).cuda()
    a = _ext.furthest_point_sampling(x, 2048)
    b = _ext.furthest_point_sampling(x, 2048)
    assert torch.equal(a, b), "FPS 80000 -> 2048: two runs differ in %d picks" % int((a != b).sum())
    ref = fx["bb_sa1_inds"].reshape(1, -1).astype(np.int32)
    got = a.cpu().numpy()
    assert np.array_equal(got, ref), "FPS 80000 -> 2048: first wrong pick at round %d" % int(np.argmax(got[0] != ref[0]))
    sub = np.ascontiguousarray(xyz[:, (seed * 997) % 70000:][:, :2048])
    c = _ext.furthest_point_sampling(torch.from_numpy(sub).cuda(), 512).cpu().numpy()
    assert np.array_equal(c, oracle.furthest_point_sampling(sub, 512)), "FPS 2048 -> 512 differs from the oracle"
    return 0.0


def check_sa_fused(seed):
    from rfdnet_amd import sa_fused
    from rfdnet_amd.pointnet2_ops.pointnet2_modules import PointnetSAModuleVotes
    worst = 0.0
    for npoint, radius, nsample, c_feat, mlp in ((256, 0.4, 32, 128, [128, 128, 128, 256]),
                                                 (512, 0.2, 64, 1, [1, 64, 64, 128])):
        mod = PointnetSAModuleVotes(npoint=npoint, radius=radius, nsample=nsample, mlp=list(mlp), use_xyz=True,
                                    normalize_xyz=True)
        synthetic.load_seeded(mod, 5)
        mod = mod.cuda().eval()
        g = torch.Generator(device="cuda").manual_seed(seed)
        xyz = (torch.rand(2, 2048, 3, device="cuda", generator=g) * 2 - 1).contiguous()
        feats = torch.randn(2, c_feat, 2048, device="cuda", generator=g)
        with torch.no_grad():
            assert sa_fused.usable(mod.mlp_module, feats, nsample, 'max', True)
            new_xyz, fused, _ = mod(xyz, feats)
            grouped, _ = mod.grouper(xyz, new_xyz, feats)
            ref = mod.mlp_module(grouped).max(dim=3)[0]
        worst = max(worst, (fused - ref).abs().max().item() / max(1.0, ref.abs().max().item()))
    assert worst < 2e-5, "fused SA layer off by %.3g" % worst
    return worst


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    assert torch.cuda.is_available()
    # cold start on purpose: the kernels under test are the FIRST launches of this context
    order = [lambda: ("gemm", check_gemm(seed)), lambda: ("decoders", check_decoders()),
             lambda: ("sa_fused", check_sa_fused(seed)), lambda: ("chain", check_chain(seed)),
             lambda: ("fps", check_fps(seed))]
    order = order[seed % 5:] + order[:seed % 5]           # rotate which kernel meets the coldest state
    if seed % 7 == 3:                                     # seeds 3, 10, 17 of the twenty
        order = [lambda: ("decoder_stress", check_decoder_stress())] + order
    res = [f() for f in order]
    _lib.device_status()
    print("fresh-process check %d OK: %s" % (seed, res))


if __name__ == "__main__":
    main()
