"""Time the split-precision GEMM at the skip-propagation encoder's shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rfdnet_amd import _lib
if os.environ.get("RFD_LIB"): _lib.LIB_PATH = os.environ["RFD_LIB"]
from rfdnet_amd import gemm

torch.manual_seed(0)
for M, N, K, res in ((262144, 512, 1536, False), (262144, 512, 1024, False), (262144, 512, 512, False),
                     (262144, 1024, 1024, False), (262144, 1024, 512, False), (262144, 512, 512, True),
                     (262144, 128, 64, False), (262144, 1024, 128, False), (262144, 512, 64, False)):
    x = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda") * 0.05
    b = torch.randn(N, device="cuda")
    r = torch.randn(M, N, device="cuda") if res else None
    out = torch.empty(M, N, device="cuda")
    for _ in range(3):
        gemm.linear(x, w, bias=b, residual=r, relu_in=True, out=out)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        gemm.linear(x, w, bias=b, residual=r, relu_in=True, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("M=%d N=%d K=%d res=%d: %.3f ms  %.1f TFLOP/s algorithmic" % (M, N, K, res, ms, 2.0 * M * N * K / ms / 1e9))
