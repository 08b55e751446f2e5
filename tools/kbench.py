"""Kernel micro-benchmarks on one MI355X (development tool, not the contract
bench).  Times each hot-path kernel with events on torch's current stream
(the stream the C-ABI launches use)."""
import json
import sys
import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rfdnet_amd import _lib, synthetic  # noqa: E402
from rfdnet_amd.pointnet2_ops import _ext  # noqa: E402
from rfdnet_amd.iscnet.occ_decoder import DecoderCBatchNorm, MODE_F16X1, MODE_F16X3  # noqa: E402


def timeit(fn, warm=2, it=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


def main():
    res = {}
    pc = synthetic.synthetic_scene(seed=10, n_points=80000)
    x = torch.from_numpy(np.ascontiguousarray(pc[None, :, :3])).cuda()
    res["fps_80000_2048_ms"] = timeit(lambda: _ext.furthest_point_sampling(x, 2048))
    inds = _ext.furthest_point_sampling(x, 2048)
    new = torch.gather(x, 1, inds.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    res["fps_2048_1024_ms"] = timeit(lambda: _ext.furthest_point_sampling(new, 1024))
    res["fps_1024_256_ms"] = timeit(lambda: _ext.furthest_point_sampling(new[:, :1024].contiguous(), 256))
    res["ballq_sa1_ms"] = timeit(lambda: _ext.ball_query(new, x, 0.2, 64))
    idx = _ext.ball_query(new, x, 0.2, 64)
    ctr = new[:, :256].contiguous()
    res["ballq_skipprop_256x80000_ns1024_ms"] = timeit(lambda: _ext.ball_query(ctr, x, 1.0, 1024))
    feats = torch.randn(1, 128, 2048, device="cuda")
    x2 = new
    c2 = new[:, :1024].contiguous()
    idx2 = _ext.ball_query(c2, x2, 0.4, 32)
    res["group_sa2_131x1024x32_ms"] = timeit(lambda: _ext.group_concat(x2, c2, feats, idx2, 0.4, True, True, True))
    # HBM-side algorithmic bytes: output written once, indices + (C,N) table + xyz read once
    # (the gathered element reads are served on chip: LDS for N <= 2048, L2 otherwise)
    nbytes = 131 * 1024 * 32 * 4 + 1024 * 32 * 4 + 128 * 2048 * 4 + 2048 * 12 + 1024 * 12
    res["group_sa2_GBps"] = nbytes / res["group_sa2_131x1024x32_ms"] / 1e6
    # batched grouping (B=32) to show the bandwidth regime
    fb = torch.randn(32, 128, 2048, device="cuda")
    xb = x2.expand(32, -1, -1).contiguous()
    cb = c2.expand(32, -1, -1).contiguous()
    ib = idx2.expand(32, -1, -1).contiguous()
    t = timeit(lambda: _ext.group_concat(xb, cb, fb, ib, 0.4, True, True, False))
    res["group_sa2_B32_ms"] = t
    res["group_sa2_B32_GBps"] = 32 * nbytes / t / 1e6

    dec = DecoderCBatchNorm(dim=3, z_dim=32, c_dim=512, hidden_size=256)
    synthetic.load_seeded(dec, 1)
    dec = dec.cuda().eval()
    K, T = 256, 32768
    p = (torch.rand(K, T, 3, device="cuda") - 0.5) * 1.1
    z = torch.zeros(K, 32, device="cuda")
    c = torch.randn(K, 512, device="cuda")
    with torch.no_grad():
        table, fcp = dec.fold(z, c)
        tile_prop = torch.arange(K, dtype=torch.int32, device="cuda").repeat_interleave(T // 128)
        pts = p.reshape(-1, 3).contiguous()
        for mode, name in ((MODE_F16X3, "f16x3"), (MODE_F16X1, "f16x1")):
            t = timeit(lambda: dec.decode_tiles(pts, tile_prop, table, fcp, mode=mode), warm=1, it=3)
            res["decode_%s_256x32768_ms" % name] = t
            res["decode_%s_TFLOPs" % name] = K * T * 1312768 / t / 1e9
            res["decode_%s_Mpts_s" % name] = K * T / t / 1e3
        res["fold_ms"] = timeit(lambda: dec.fold(z, c))
        # SURVEY config 3: decoder stress, 256 x 262 144 uniform points (88.1 TFLOP algorithmic)
        T3 = 262144
        p3 = (torch.rand(K * T3, 3, device="cuda") - 0.5) * 1.1
        tp3 = torch.arange(K, dtype=torch.int32, device="cuda").repeat_interleave(T3 // 128)
        for mode, name in ((MODE_F16X3, "f16x3"), (MODE_F16X1, "f16x1")):
            t = timeit(lambda: dec.decode_tiles(p3, tp3, table, fcp, mode=mode), warm=1, it=2)
            res["config3_decode_%s_256x262144_ms" % name] = t
            res["config3_decode_%s_TFLOPs" % name] = K * T3 * 1312768 / t / 1e9
            res["config3_decode_%s_frac_of_2500" % name] = K * T3 * 1312768 / t / 1e9 / 2500.0
        del p3, tp3
    res["fps_sa1_Mupdates_per_s"] = 80000 * 2047 / res["fps_80000_2048_ms"] / 1e3
    res["fps_sa1_onchip_GBps"] = 16.0 * 80000 * 2047 / res["fps_80000_2048_ms"] / 1e6
    res["ballq_sa1_Gtests_per_s_upper"] = 2048 * 80000 / res["ballq_sa1_ms"] / 1e6
    res["group_sa2_B32_frac_of_8TBps"] = res["group_sa2_B32_GBps"] / 8000.0
    # Chamfer nearest-neighbour search at fit_mesh_to_scan sizes (network.py:194-195)
    from rfdnet_amd import chamfer_distance as cdm
    Bc, nc, mc = 16, 10000, 50000
    a1 = torch.randn(Bc, nc, 3, device="cuda")
    a2 = torch.randn(Bc, mc, 3, device="cuda")
    t = timeit(lambda: cdm.nearest(a1, a2), warm=1, it=3)
    res["chamfer_16x10000x50000_ms"] = t
    res["chamfer_Gpairs_per_s"] = 2.0 * Bc * nc * mc / t / 1e6
    res["chamfer_valu_TFLOPs_8flop_per_pair"] = 8 * 2.0 * Bc * nc * mc / t / 1e9
    del a1, a2
    from rfdnet_amd import gemm
    for (M, N, K) in ((262144, 1024, 1024), (262144, 1024, 512), (262144, 512, 512)):
        xa = torch.randn(M, K, device="cuda")
        wa = torch.randn(N, K, device="cuda") / K ** 0.5
        oa = torch.empty(M, N, device="cuda")
        t = timeit(lambda: gemm.linear(xa, wa, relu_in=True, out=oa), warm=1, it=3)
        res["gemm_f16x3_%dx%dx%d_ms" % (M, N, K)] = t
        res["gemm_f16x3_%dx%dx%d_TFLOPs" % (M, N, K)] = 2.0 * M * N * K / t / 1e9
        t2 = timeit(lambda: torch.mm(torch.relu(xa), wa.t(), out=oa), warm=1, it=3)
        res["torch_fp32_%dx%dx%d_ms" % (M, N, K)] = t2
        del xa, wa, oa
    _lib.device_status()
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
