import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from rfdnet_amd import synthetic
from rfdnet_amd.iscnet.config import Config
from rfdnet_amd.iscnet.network import ISCNet
import rfdnet_amd.iscnet.skip_propagation as SP
cfg = Config({'generation': {'resolution_0': 32, 'upsampling_steps': 1}})
net = ISCNet(cfg); synthetic.load_seeded(net, 10); net = net.cuda().eval()
pc = torch.from_numpy(synthetic.synthetic_scene(seed=10, n_points=80000)[None]).cuda()
orig = net.skip_propagation.point_seg.forward_rows
def spy(inp):
    out = orig(inp)
    m = torch.argmax(out[0].view(-1, 2), dim=1).view(inp.shape[0], inp.shape[1])
    k = m.sum(1).cpu().numpy()
    print("kept per proposal: mean %.1f min %d max %d of %d; padded-to-256 rows: %d of %d" % (k.mean(), k.min(), k.max(), inp.shape[1], int((np.ceil((k + 1) / 256) * 256).sum()), inp.shape[0] * inp.shape[1]))
    # duplicates among the grouped points (ball query pads with the first hit)
    x = inp[:, :, :3]
    return out
net.skip_propagation.point_seg.forward_rows = spy
with torch.no_grad():
    ep, pf = net.detect(pc)
    ids = net.select_proposals(ep, 'all', pc)
    net.object_codes(ep, pf, ids, pc)
