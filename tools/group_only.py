"""Run only the fused grouping kernel (SA2 shape) at B = 32 and B = 8 scenes per launch for rocprofv3 counter
collection (tools/pmc_traffic.py --group reads the two passes)."""
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
for B in (32, 8):            # dispatch order: 5 launches at B = 32, then 5 at B = 8
    fb = torch.randn(B, 128, 2048, device="cuda")
    xb = x2.expand(B, -1, -1).contiguous(); cb = c2.expand(B, -1, -1).contiguous(); ib = idx2.expand(B, -1, -1).contiguous()
    for _ in range(5):
        _ext.group_concat(xb, cb, fb, ib, 0.4, True, True, False)
    torch.cuda.synchronize()
