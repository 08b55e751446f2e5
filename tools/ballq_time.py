"""Time the two 80 000-point ball queries of a scene (SA1 and skip propagation)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rfdnet_amd import synthetic
from rfdnet_amd.pointnet2_ops import _ext
pc = synthetic.synthetic_scene(seed=10, n_points=80000)
x = torch.from_numpy(np.ascontiguousarray(pc[None, :, :3])).cuda()
inds = _ext.furthest_point_sampling(x, 2048)
new = torch.gather(x, 1, inds.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
ctr = new[:, :256].contiguous()
def t(f, n=30):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
print("SA1  2048 x 80000, r=0.2, ns=64   : %.1f us" % t(lambda: _ext.ball_query(new, x, 0.2, 64)))
print("skip  256 x 80000, r=1.0, ns=1024 : %.1f us" % t(lambda: _ext.ball_query(ctr, x, 1.0, 1024)))
print("SA2  1024 x 2048,  r=0.4, ns=32   : %.1f us" % t(lambda: _ext.ball_query(new[:, :1024].contiguous(), new, 0.4, 32)))
