"""Tail launches of the occupancy decoder: the main kernel (a workgroup per CU, 155 KiB of LDS) against csrc/occ_decoder_tail.hip
(one wave per 16 points, no LDS) on launch shapes of the last MISE rounds -- alone, and while another stream keeps every CU busy with
small workgroups (what the other scenes in flight do to a tail launch in the pipeline).  Prints one table; GPU only.

    python tools/decoder_tail_bench.py [--reps 50]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rfdnet_amd import _lib, synthetic  # noqa: E402
from rfdnet_amd.iscnet.occ_decoder import DecoderCBatchNorm  # noqa: E402


def case(dec, K, tiles_per_prop, real_per_tile, seed=0):
    rng = np.random.default_rng(seed)
    tile_prop = np.repeat(np.arange(K, dtype=np.int32), tiles_per_prop)
    n = tile_prop.shape[0] * 128
    g = torch.Generator(device="cuda").manual_seed(seed)
    pts = ((torch.rand(n, 3, device="cuda", generator=g) - 0.5) * 1.1).contiguous()
    c = torch.randn(K, 512, device="cuda", generator=g)
    with torch.no_grad():
        table, fcp = dec.fold(torch.zeros(K, 32, device="cuda"), c)
    lin = np.full(n, -1, dtype=np.int32)
    for t in range(tile_prop.shape[0]):
        r = min(128, max(1, int(rng.poisson(real_per_tile))))
        lin[t * 128:t * 128 + r] = np.arange(r) + (t % tiles_per_prop) * 128          # real slots lead a tile, as the tile builder packs them
    values = torch.zeros(K, tiles_per_prop * 128, device="cuda")
    pstate = torch.ones(K, tiles_per_prop * 128, dtype=torch.uint8, device="cuda")
    return pts, torch.from_numpy(tile_prop).cuda(), table, fcp, (torch.from_numpy(lin).cuda(), values, pstate)


def timed(fn, reps, busy=None):
    """mean ms per call, HIP events on the launch stream; busy = a callable that keeps another stream full meanwhile"""
    fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        if busy is not None:
            with torch.cuda.stream(side):
                busy()
        a.record()
        fn()
        b.record()
        if busy is not None:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    return sum(t) / len(t), t[len(t) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=50)
    a = ap.parse_args()
    dec = DecoderCBatchNorm(dim=3, z_dim=32, c_dim=512, hidden_size=256)
    synthetic.load_seeded(dec, 1)
    dec = dec.cuda().eval()
    lib = _lib.lib()
    big = torch.empty(64 << 20, device="cuda")           # 256 MB: an HBM-bound elementwise pass of ~0.15 ms, 64 K small workgroups

    def busy():
        for _ in range(12):
            big.mul_(1.0001)

    print("# K proposals x tiles per proposal, ~real points per tile | main kernel alone / busy | tail kernel alone / busy   (mean ms, median ms)")
    for K, tpp, real in ((256, 1, 18), (128, 1, 18), (256, 1, 70), (104, 1, 12), (64, 4, 128), (256, 2, 128)):
        pts, tile_prop, table, fcp, sc = case(dec, K, tpp, real)
        row = []
        for tail in (0, 100000):
            lib.rfd_occ_set_tail_tiles(tail)
            for b in (None, busy):
                with torch.no_grad():
                    row.append(timed(lambda: dec.decode_tiles(pts, tile_prop, table, fcp, scatter=sc), a.reps, b))
        lib.rfd_occ_set_tail_tiles(384)
        print("%4d x %d, %3d real/tile (%5d tiles, %6d real points) | %6.3f (%6.3f) / %6.3f (%6.3f) | %6.3f (%6.3f) / %6.3f (%6.3f)" % (
            K, tpp, real, tile_prop.shape[0], int((sc[0] >= 0).sum()), *[x for r in row for x in r]))


if __name__ == "__main__":
    main()
