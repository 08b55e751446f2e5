"""Stage-level GPU time of one scene (events on the current stream)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rfdnet_amd import synthetic
from rfdnet_amd.iscnet.config import Config
from rfdnet_amd.iscnet.network import ISCNet

import os
cfg = Config({'data': {'num_point': int(os.environ.get('ST_POINTS', 80000))}, 'generation': {'resolution_0': 32, 'upsampling_steps': int(os.environ.get('ST_STEPS', 1))}})
net = ISCNet(cfg); synthetic.load_seeded(net, 10); net = net.cuda().eval()
pc = torch.from_numpy(synthetic.synthetic_scene(seed=10, n_points=int(os.environ.get('ST_POINTS', 80000)))[None]).cuda()


class T:
    def __init__(self): self.ev = []
    def mark(self, name):
        e = torch.cuda.Event(enable_timing=True); e.record(); self.ev.append((name, e, time.perf_counter()))
    def report(self):
        torch.cuda.synchronize()
        for (n0, e0, h0), (n1, e1, h1) in zip(self.ev, self.ev[1:]):
            print("  %-28s gpu %8.3f ms   host %8.3f ms" % (n1, e0.elapsed_time(e1), (h1 - h0) * 1e3))
        print("  total gpu %.3f ms" % self.ev[0][1].elapsed_time(self.ev[-1][1]))


with torch.no_grad():
    for rep in range(3):
        t = T(); t.mark("start")
        ep = net.backbone(pc, {}); t.mark("backbone")
        xyz, feats = ep['fp2_xyz'], ep['fp2_features']
        ep['seed_inds'] = ep['fp2_inds']; ep['seed_xyz'] = xyz; ep['seed_features'] = feats
        vx, vf = net.voting(xyz, feats); vf = vf.div(torch.norm(vf, p=2, dim=1).unsqueeze(1)); t.mark("voting")
        ep, pf = net.detection(vx, vf, ep, True); t.mark("proposal")
        ids = net.select_proposals(ep, 'all'); t.mark("select(all)")
        codes = net.object_codes(ep, pf, ids, pc); t.mark("skip_propagation")
        cls = net.cls_codes(ep, ids)
        gen = net.completion.generator
        grids = gen.generate_grids(codes, cls); t.mark("decode+MISE")
        meshes = gen.extract_meshes(grids); t.mark("marching_cubes")
        v = torch.cat([m.vertices for m in meshes]).cpu(); f = torch.cat([m.faces for m in meshes]).cpu(); t.mark("meshes->host")
        if rep == 2:
            t.report(); print(gen.stats, v.shape, f.shape)
    # backbone detail
    t = T(); t.mark("start")
    x, fe = net.backbone._break_up_pc(pc)
    for i in (1, 2, 3, 4):
        x, fe, _ = getattr(net.backbone, 'sa%d' % i)(x, fe); t.mark("sa%d" % i)
    t.report()
