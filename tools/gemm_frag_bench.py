"""Encoder GEMM pair, fp32-rows kernels (round 1-5) against the frag-rows kernels (round 6), at the headline shapes.

    python tools/gemm_frag_bench.py            -> profiles/r06_gemm_frag_bench.txt

One ResnetBlockFC of the skip-propagation encoder = G1 (fc_0: M x 512 x K1) + G2 ([fc_1 | shortcut]: M x 512 x K2),
M = 256 proposals x 1024 points; block 0: K1 = 1024, K2 = 1536; blocks 1-4: K1 = 512, K2 = 1024.  Also fc_pos."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from rfdnet_amd import gemm, pos_embed  # noqa: E402


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    torch.manual_seed(0)
    M, h, T = 262144, 512, 1024
    sa = gemm.SA
    total_rows = total_frag = 0.0
    for name, kin in (("block 0", 2 * h), ("blocks 1-4", h)):
        w1 = torch.randn(h, kin, device="cuda") * 0.05
        w2 = torch.randn(h, h + kin, device="cuda") * 0.05
        gb = torch.randn(M // T, h, device="cuda")
        cat = torch.randn(M, h + kin, device="cuda")
        nxt = torch.empty(M, 2 * h, device="cuda")
        pool = torch.zeros(M // T, h, device="cuda")
        t1 = timed(lambda: gemm.linear(cat[:, h:], w1, gbias=gb, rows_per_group=T, relu_in=True, out=cat[:, :h]))
        t2 = timed(lambda: gemm.linear(cat, w2, gbias=gb, rows_per_group=T, relu_in=True, out=nxt[:, h:], pool=pool))
        t2p = timed(lambda: gemm.linear(cat, w2, gbias=gb, rows_per_group=T, relu_in=True, pool=pool, store=False))
        fcat = gemm.rows_to_frag(cat, sa=sa)
        fnxt = gemm.frag_empty(M, 2 * h, "cuda")
        hb = h // 32
        f1 = timed(lambda: gemm.linear_frag(fcat[:, hb:], w1, gbias=gb, rows_per_group=T, out=fcat[:, :hb], sa=sa))
        f2 = timed(lambda: gemm.linear_frag(fcat, w2, gbias=gb, rows_per_group=T, out=fnxt[:, hb:], pool=pool, sa=sa))
        f2p = timed(lambda: gemm.linear_frag(fcat, w2, gbias=gb, rows_per_group=T, pool=pool, store=False, sa=sa))
        fl1, fl2 = 2.0 * M * h * kin, 2.0 * M * h * (h + kin)
        print("%s (K1 = %d, K2 = %d)" % (name, kin, h + kin))
        for tag, a, b, fl in (("G1 fc_0", t1, f1, fl1), ("G2 [fc_1|shortcut] + pool", t2, f2, fl2),
                              ("G2 pool only (last block)", t2p, f2p, fl2)):
            print("   %-28s fp32 rows %.3f ms (%.0f TF)   frag rows %.3f ms (%.0f TF)   x%.2f" % (
                tag, a, fl / a / 1e9, b, fl / b / 1e9, a / b))
        n = 1 if kin == 2 * h else 3
        total_rows += n * (t1 + t2) + (0 if kin == 2 * h else t1 + t2p)
        total_frag += n * (f1 + f2) + (0 if kin == 2 * h else f1 + f2p)
    print("ten GEMMs of one scene: fp32 rows %.2f ms, frag rows %.2f ms" % (total_rows, total_frag))
    # fc_pos
    d = 4
    x = torch.randn(M, d, device="cuda")
    mask = (torch.rand(M, device="cuda") > 0.2).float()
    W = torch.randn(2 * h, d + 128, device="cuda") * 0.3
    bias = torch.randn(2 * h, device="cuda")
    group = torch.randn(M // T, 2 * h, device="cuda")
    out = torch.empty(M, 3 * h, device="cuda")
    fout = gemm.frag_empty(M, 3 * h, "cuda")
    p0 = timed(lambda: pos_embed.pos_embed(x, mask, W, bias, group, T, out[:, h:]))
    p1 = timed(lambda: pos_embed.pos_embed_frag(x, mask, W, bias, group, T, fout[:, h // 32:], sa))
    print("fc_pos (M x 1024): fp32 rows %.3f ms   frag rows %.3f ms" % (p0, p1))


if __name__ == "__main__":
    main()
