"""Time the fused grouping kernel (SA2 shape) at B = 1 and B = 32; A/B two libraries with RFD_HIP_LIB."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rfdnet_amd import synthetic
from rfdnet_amd.pointnet2_ops import _ext
pc = synthetic.synthetic_scene(seed=10, n_points=80000)
x = torch.from_numpy(np.ascontiguousarray(pc[None, :, :3])).cuda()
inds = _ext.furthest_point_sampling(x, 2048)
x2 = torch.gather(x, 1, inds.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
c2 = x2[:, :1024].contiguous()
idx2 = _ext.ball_query(c2, x2, 0.4, 32)
def t(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for B in (1, 32):
    fb = torch.randn(B, 128, 2048, device="cuda")
    xb = x2.expand(B, -1, -1).contiguous(); cb = c2.expand(B, -1, -1).contiguous(); ib = idx2.expand(B, -1, -1).contiguous()
    us = t(lambda: _ext.group_concat(xb, cb, fb, ib, 0.4, True, True, False))
    wr = B * 131 * 1024 * 32 * 4
    print("B=%d  %.1f us   written %.0f MB  -> %.0f GB/s of writes" % (B, us, wr / 1e6, wr / us / 1e3))
    us = t(lambda: _ext.group_points(fb, ib))
    print("   group_points %.1f us" % us)
