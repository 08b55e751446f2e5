import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rfdnet_amd import gemm
torch.manual_seed(0)
M, N, K = 262144, 1024, 512
x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.05; b = torch.randn(N, device="cuda")
out = torch.empty(M, N, device="cuda")
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    gemm.linear(x, w, bias=b, relu_in=True, out=out)
torch.cuda.synchronize()
