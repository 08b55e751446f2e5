"""Debug: phase stamps (s_memtime) of FPS rounds 100..103, workgroup 0."""
import ctypes as C, os, subprocess, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rfdnet_amd import _lib, build, synthetic
# the stamps are not in the product source: tools/ab/fps_trace.patch puts them into a scratch copy of sampling.hip
sys.path.insert(0, os.path.join(ROOT, "tools", "ab"))
import build_variants  # noqa: E402
so = os.path.join(ROOT, "rfdnet_amd", "lib", "variants", "librfd_fps_trace.so")
if not os.path.exists(so) or "--build-only" in sys.argv or "--rebuild" in sys.argv:
    build_variants.build_patched(so, "sampling.hip", "fps_trace.patch", ["-DRFD_FPS_TRACE"])
if "--build-only" in sys.argv:
    sys.exit(0)
_lib.LIB_PATH = so
lib = _lib.lib()
for n, m in ((80000, 2048), (2048, 1024)):
    pc = synthetic.synthetic_scene(seed=10, n_points=80000)[:n]
    x = torch.from_numpy(np.ascontiguousarray(pc[None, :, :3])).cuda()
    tmp = torch.zeros(1, max(n, 4096), device="cuda")
    out = torch.zeros(1, m, dtype=torch.int32, device="cuda")
    for _ in range(2):
        lib.furthest_point_sampling_kernel_wrapper(1, n, m, x.data_ptr(), tmp.data_ptr(), out.data_ptr(), _lib.current_stream())
    torch.cuda.synchronize()
    st = tmp.cpu().numpy().view(np.uint64)[0][:32].reshape(4, 8).astype(np.int64)
    print("n=%d m=%d" % (n, m))
    for r in range(3):
        d = np.diff(st[r][:7])
        print("  round %d: update %d  wave_sel %d  lds+bar %d  publish %d  gather %d  select2 %d | round total %d"
              % (100 + r, d[0], d[1], d[2], d[3], d[4], d[5], st[r + 1][0] - st[r][0]))
