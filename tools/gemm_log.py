"""List every split-precision GEMM of one skip-propagation pass with its shape and time."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rfdnet_amd import synthetic, gemm
from rfdnet_amd.iscnet.config import Config
from rfdnet_amd.iscnet.network import ISCNet

cfg = Config({'data': {'num_point': 80000}, 'generation': {'resolution_0': 32, 'upsampling_steps': 1}})
net = ISCNet(cfg); synthetic.load_seeded(net, 10); net = net.cuda().eval()
pc = torch.from_numpy(synthetic.synthetic_scene(seed=10, n_points=80000)[None]).cuda()
log = []
orig = gemm.linear
def spy(x, weight, **kw):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); out = orig(x, weight, **kw); b.record()
    log.append((x.shape[0], weight.shape[0], x.shape[1], x.stride(0), kw.get('relu_in', False), kw.get('gbias') is not None,
                kw.get('residual') is not None, kw.get('pool') is not None, kw.get('store', True), a, b))
    return out
with torch.no_grad():
    for rep in range(3):
        ep = net.backbone(pc, {})
        xyz, feats = ep['fp2_xyz'], ep['fp2_features']
        ep['seed_inds'] = ep['fp2_inds']; ep['seed_xyz'] = xyz; ep['seed_features'] = feats
        vx, vf = net.voting(xyz, feats); vf = vf.div(torch.norm(vf, p=2, dim=1).unsqueeze(1))
        ep, pf = net.detection(vx, vf, ep, True)
        ids = net.select_proposals(ep, 'all')
        if rep == 2:
            gemm.linear = spy
        codes = net.object_codes(ep, pf, ids, pc)
torch.cuda.synchronize()
tot = 0.0
for M, N, K, lda, ri, gb, res, pool, store, a, b in log:
    ms = a.elapsed_time(b); tot += ms
    print("M=%7d N=%5d K=%5d lda=%5d relu_in=%d gbias=%d res=%d pool=%d store=%d  %.3f ms  %.0f TF" %
          (M, N, K, lda, ri, gb, res, pool, store, ms, 2.0 * M * N * K / ms / 1e9))
print("total %.3f ms over %d GEMMs" % (tot, len(log)))
