"""Time the fused ResnetBlockFC kernel against the two split-precision GEMMs it
replaces (skip-propagation shapes: 256 proposals x 2048 points)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rfdnet_amd import resblock, gemm
from rfdnet_amd.iscnet.layers import ResnetBlockFC

torch.manual_seed(0)
blk = ResnetBlockFC(512, 256).cuda()
with torch.no_grad():
    blk.fc_1.weight.copy_(torch.randn(256, 256, device="cuda") * 0.06)
for k_in, G, T in ((256, 256, 2048), (512, 1, 256 * 2048)):
    M = G * T
    x = torch.randn(M, k_in, device="cuda")
    g0 = torch.randn(G, 256, device="cuda"); gs = torch.randn(G, 256, device="cuda")
    out = torch.empty(M, 256, device="cuda")
    for _ in range(3):
        resblock.forward(blk, x, g0, gs, T, out=out)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        resblock.forward(blk, x, g0, gs, T, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    flop = 2.0 * M * (2 * k_in * 256 + 256 * 256)
    print("k_in=%d M=%d: %.3f ms  %.1f TFLOP/s algorithmic (x3 issued)  %.2f TB/s" %
          (k_in, M, ms, flop / ms / 1e9, (M * (k_in + 256) * 4) / ms / 1e9))
