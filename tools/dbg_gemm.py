import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from rfdnet_amd import gemm
torch.manual_seed(0)
M, N, K = 256, 256, int(sys.argv[1]) if len(sys.argv) > 1 else 128
x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.1
y = gemm.linear(x, w)
r = (x.double() @ w.double().t())
e = (y.double() - r).abs()
print("max err", e.max().item())
bad = (e > 1e-3)
print("bad fraction", bad.float().mean().item())
print("bad rows", bad.any(1).nonzero().flatten()[:20].tolist(), "bad cols", bad.any(0).nonzero().flatten()[:20].tolist())
# per-k contribution test: which k indices are wrong? use one-hot x
for kk in (0, 7, 8, 15, 16, 31, 32, 33, 64, 96, 127, K - 1):
    x1 = torch.zeros(M, K, device="cuda"); x1[:, kk] = 1.0
    y1 = gemm.linear(x1, w)
    print("k", kk, "err", (y1 - w[:, kk][None, :]).abs().max().item())
