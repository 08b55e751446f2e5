import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from rfdnet_amd import synthetic, _lib
from rfdnet_amd.iscnet.occ_decoder import DecoderCBatchNorm
dec = DecoderCBatchNorm(dim=3, z_dim=32, c_dim=512, hidden_size=256); synthetic.load_seeded(dec, 99); dec = dec.cuda().eval()
rng = np.random.default_rng(5)
K, T = 8, 1024
p = torch.from_numpy(((rng.random((K, T, 3)) - 0.5) * 1.1).astype(np.float32)).cuda()
z = torch.from_numpy(rng.normal(0, 1, (K, 32)).astype(np.float32)).cuda()
c = torch.from_numpy(rng.normal(0, 1, (K, 512)).astype(np.float32)).cuda()
with torch.no_grad():
    outs = [dec(p, z, c).cpu().numpy() for _ in range(8)]
ref = np.median(np.stack(outs), axis=0)          # majority value per point
if os.environ.get("RFD_DBG_DUMP"):               # the eight launches' logits, for tools/fault_model.py
    os.makedirs(os.environ["RFD_DBG_DUMP"], exist_ok=True)
    np.save(os.path.join(os.environ["RFD_DBG_DUMP"], "outs_%d.npy" % os.getpid()), np.stack(outs))
for r, o in enumerate(outs):
    bad = np.abs(o - ref) > 1e-5
    if bad.any():
        flat = np.argwhere(bad)
        tiles = {}
        for k, t in flat:
            tiles.setdefault((int(k), int(t) // 128), []).append(int(t) % 128)
        desc = []
        for (k, tl), pts in sorted(tiles.items()):
            waves = sorted(set(x // 16 for x in pts))
            desc.append("prop %d tile %d: %d pts, waves %s, max err %.2e" % (k, tl, len(pts), waves, np.abs(o - ref)[k, tl*128:(tl+1)*128].max()))
        print("run %d BAD: %s" % (r, "; ".join(desc)))
    else:
        print("run %d ok" % r)
