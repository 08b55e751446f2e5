import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rfdnet_amd import synthetic
from rfdnet_amd.iscnet.config import Config
from rfdnet_amd.iscnet.network import ISCNet
from rfdnet_amd.iscnet import skip_propagation as sp
cfg = Config({'generation': {'resolution_0': 32, 'upsampling_steps': 1}})
net = ISCNet(cfg); synthetic.load_seeded(net, 10); net = net.cuda().eval()
orig_argmax = torch.argmax
def dbg_argmax(x, dim=None, **k):
    r = orig_argmax(x, dim=dim, **k)
    if x.dim() == 2 and x.shape[1] == 2:
        m = r.view(256, -1).float()
        print("mask fraction %.4f  per-proposal min %.3f max %.3f  shape %s" % (m.mean().item(), m.mean(1).min().item(), m.mean(1).max().item(), tuple(m.shape)))
    return r
torch.argmax = dbg_argmax
for seed in (10, 11):
    pc = torch.from_numpy(synthetic.synthetic_scene(seed=seed, n_points=80000)[None]).cuda()
    with torch.no_grad():
        ep, pf = net.detect(pc)
        ids = net.select_proposals(ep, 'all', pc)
        codes = net.object_codes(ep, pf, ids, pc)
