"""Timing-only ablations of gemm_rowsf_kernel (tools/ab/gemm_frag_ablation.patch; results of the variants are WRONG):
what bounds the frag-rows encoder GEMM?   python tools/ab/gemm_frag_ab.py [variant ...]
Variants are librfd_<name>.so under rfdnet_amd/lib/variants/ (built from the patch with -DAB_* flags)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def child():
    import torch
    from rfdnet_amd import _lib
    _lib.LIB_PATH = os.environ["RFD_LIB"]
    from rfdnet_amd import gemm
    torch.manual_seed(0)
    M, h, T = 262144, 512, 1024
    sa = gemm.SA
    out = []
    for kin in (h, 2 * h):
        w1 = torch.randn(h, kin, device="cuda") * 0.05
        w2 = torch.randn(h, h + kin, device="cuda") * 0.05
        gb = torch.randn(M // T, h, device="cuda")
        fcat = gemm.rows_to_frag(torch.randn(M, h + kin, device="cuda"), sa=sa)
        fnxt = gemm.frag_empty(M, 2 * h, "cuda")
        pool = torch.zeros(M // T, h, device="cuda")
        hb = h // 32

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
        f1 = timed(lambda: gemm.linear_frag(fcat[:, hb:], w1, gbias=gb, rows_per_group=T, out=fcat[:, :hb], sa=sa))
        f2 = timed(lambda: gemm.linear_frag(fcat, w2, gbias=gb, rows_per_group=T, out=fnxt[:, hb:], pool=pool, sa=sa))
        out.append("K1=%d %.3f ms (%.0f TF) | K2=%d %.3f ms (%.0f TF)" % (
            kin, f1, 2.0 * M * h * kin / f1 / 1e9, h + kin, f2, 2.0 * M * h * (h + kin) / f2 / 1e9))
    print("%-14s %s" % (os.path.basename(os.environ["RFD_LIB"])[7:-3], "  ||  ".join(out)))


if __name__ == "__main__":
    if os.environ.get("RFD_AB_CHILD"):
        child()
    else:
        vdir = os.path.join(ROOT, "rfdnet_amd", "lib", "variants")
        names = sys.argv[1:] or sorted(f[7:-3] for f in os.listdir(vdir) if f.startswith("librfd_gf_"))
        for n in names:
            env = dict(os.environ, RFD_AB_CHILD="1", RFD_LIB=os.path.join(vdir, "librfd_%s.so" % n))
            subprocess.run([sys.executable, os.path.abspath(__file__)], env=env)
