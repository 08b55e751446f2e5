"""FPS exchange-geometry sweep: SA1 shape (80 000 -> 2048), G workgroups x 256 threads x
PPT points per thread for PPT in {5, 8, 10, 16, 20, 40} (G = 63 .. 8), every geometry checked bit-equal to the default
one's indices before it is timed (HIP events on the launch stream, 7 launches after 2 warm-ups, median).
  python tools/fps_sweep.py > profiles/r05_fps_sweep.txt      (one GPU)
Also times the single-workgroup levels of the backbone (2048 -> 1024, 1024 -> 512, 512 -> 256) for the record."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rfdnet_amd import _lib, synthetic  # noqa: E402
from rfdnet_amd.pointnet2_ops import _ext  # noqa: E402


def timed(fn, it=7):
    for _ in range(2):
        fn()
    ms = []
    for _ in range(it):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        ms.append(e0.elapsed_time(e1))
    return float(np.median(ms)), float(np.min(ms))


def main():
    lib = _lib.lib()
    pc = synthetic.synthetic_scene(seed=10, n_points=80000)
    x = torch.from_numpy(np.ascontiguousarray(pc[None, :, :3])).cuda()
    N, M = 80000, 2048
    ref = _ext.furthest_point_sampling(x, M)
    _lib.device_status()
    print("FPS geometry sweep, SA1 shape %d -> %d, synthetic scene seed 10; %s" % (N, M, torch.cuda.get_device_name(0)))
    print("%4s %4s %10s %10s %12s %s" % ("PPT", "G", "median ms", "min ms", "us / round", "bit-equal to default"))
    rows = []
    for ppt in (0, 5, 8, 10, 16, 20, 40):
        assert lib.rfd_fps_set_geometry(ppt) >= 0
        try:
            out = _ext.furthest_point_sampling(x, M)
            _lib.device_status()
            same = bool(torch.equal(out, ref))
            med, mn = timed(lambda: _ext.furthest_point_sampling(x, M))
        finally:
            lib.rfd_fps_set_geometry(0)
        p = ppt or 10
        G = (N + 256 * p - 1) // (256 * p)
        rows.append((ppt, G, med, mn, same))
        print("%4s %4d %10.3f %10.3f %12.3f %s" % (ppt or "def", G, med, mn, 1e3 * med / (M - 1), same))
    sub = x
    print("single-workgroup levels (register-resident, no exchange):")
    for n, m in ((2048, 1024), (1024, 512), (512, 256), (1024, 256)):
        sub = x[:, :n].contiguous()
        med, mn = timed(lambda: _ext.furthest_point_sampling(sub, m))
        print("  %5d -> %4d  %8.3f ms  %8.3f us / round" % (n, m, med, 1e3 * med / (m - 1)))
    base = [r for r in rows if r[0] == 0][0][2]
    best = min(rows[1:], key=lambda r: r[2])
    print("default %.3f ms; best forced geometry PPT %d (G %d) %.3f ms = %+.1f %%"
          % (base, best[0], best[1], best[2], 100.0 * (best[2] / base - 1.0)))


if __name__ == "__main__":
    main()
