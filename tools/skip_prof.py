"""torch.profiler view of the skip-propagation stage (which host op launches which kernel)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rfdnet_amd import synthetic
from rfdnet_amd.iscnet.config import Config
from rfdnet_amd.iscnet.network import ISCNet
from torch.profiler import profile, ProfilerActivity

cfg = Config({'data': {'num_point': 80000}, 'generation': {'resolution_0': 32, 'upsampling_steps': 1}})
net = ISCNet(cfg); synthetic.load_seeded(net, 10); net = net.cuda().eval()
pc = torch.from_numpy(synthetic.synthetic_scene(seed=10, n_points=80000)[None]).cuda()
with torch.no_grad():
    for rep in range(3):
        ep = net.backbone(pc, {})
        xyz, feats = ep['fp2_xyz'], ep['fp2_features']
        ep['seed_inds'] = ep['fp2_inds']; ep['seed_xyz'] = xyz; ep['seed_features'] = feats
        vx, vf = net.voting(xyz, feats); vf = vf.div(torch.norm(vf, p=2, dim=1).unsqueeze(1))
        ep, pf = net.detection(vx, vf, ep, True)
        ids = net.select_proposals(ep, 'all')
        torch.cuda.synchronize()
        if rep == 2:
            with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
                codes = net.object_codes(ep, pf, ids, pc)
                torch.cuda.synchronize()
            print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=60))
            print(prof.key_averages(group_by_stack_n=6).table(sort_by="cuda_time_total", row_limit=12, max_name_column_width=50))
        else:
            codes = net.object_codes(ep, pf, ids, pc)
