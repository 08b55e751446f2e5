"""s_memtime trace of the row-owner GEMM (side build with -DRFD_GEMM_TRACE)."""
import sys, os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from rfdnet_amd import _lib, build
so = os.path.join(ROOT, "rfdnet_amd", "lib", "librfd_hip_trace.so")
if not os.path.exists(so) or "--rebuild" in sys.argv:
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + build.HIPCC_FLAGS + ["-DRFD_GEMM_TRACE", "-o", so] + build.sources())
if "--build-only" in sys.argv:
    sys.exit(0)
_lib.LIB_PATH = so
from rfdnet_amd import gemm
torch.manual_seed(0)
M, N, K = 262144, 1024, 512
x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.05
full = torch.zeros(M + 64, N, device="cuda")          # stamps land behind the M rows
out = full[:M]
for _ in range(2):
    gemm.linear(x, w, relu_in=True, out=out)
torch.cuda.synchronize()
raw = full[M:].reshape(-1)[:64 * 256].cpu().numpy().view(np.uint64).reshape(64, 128)
np.set_printoptions(linewidth=200)
for wg in (0, 1, 5, 20, 40, 63):
    t = raw[wg].astype(np.int64)
    n = int((t != 0).sum())
    print("wg", wg * 8, "stamps", n, "prologue", t[1] - t[0])
    st = t[2:2 + 4 * (K // 32)].reshape(-1, 4)
    d = np.diff(np.concatenate([st.reshape(-1), [t[2 + 4 * (K // 32)]]])).reshape(-1, 4)
    print("  per step [s0 loop, wait8, s1 loop, barrier]:")
    print(d[:16])
    print("  mean", d.mean(0), "step total", d.sum(1).mean())
    end_loop = t[2 + 4 * (K // 32)]
    print("  whole WG", t[127] - t[0], " loop", end_loop - t[2], " epilogue", t[127] - end_loop)
