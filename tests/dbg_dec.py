"""Multi-proposal decoder check against the CPU oracle, one fresh process (flakiness hunt)."""
import os, sys, numpy as np, torch
from collections import OrderedDict
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from rfdnet_amd import synthetic, _lib
from oracle import oracle
from rfdnet_amd.iscnet.occ_decoder import DecoderCBatchNorm
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
dec = DecoderCBatchNorm(dim=3, z_dim=32, c_dim=512, hidden_size=256); synthetic.load_seeded(dec, 99); dec = dec.cuda().eval()
sd = OrderedDict((k, v.detach().cpu().numpy()) for k, v in dec.state_dict().items())
blob = oracle.decoder_param_blob(sd)
rng = np.random.default_rng(5)
worst = 0.0
bad = []
for K, T in ((5, 333), (3, 517), (8, 1024)):
    p = ((rng.random((K, T, 3)) - 0.5) * 1.1).astype(np.float32)
    z = rng.normal(0, 1, (K, 32)).astype(np.float32)
    c = rng.normal(0, 1, (K, 512)).astype(np.float32)
    ref = oracle.decoder_cbn(blob, p, z, c)
    ref2 = oracle.decoder_cbn(blob, p, z, c)
    assert np.array_equal(ref, ref2), "ORACLE not deterministic"
    with torch.no_grad():
        out = dec(torch.from_numpy(p).cuda(), torch.from_numpy(z).cuda(), torch.from_numpy(c).cuda()).cpu().numpy()
        out2 = dec(torch.from_numpy(p).cuda(), torch.from_numpy(z).cuda(), torch.from_numpy(c).cuda()).cpu().numpy()
    e = np.abs(out - ref).max()
    worst = max(worst, e)
    if e > 1e-5 or not np.array_equal(out, out2):
        bad.append((K, T, float(e), float(np.abs(out - out2).max()), np.argwhere(np.abs(out - ref) > 1e-5)[:6].tolist()))
_lib.device_status()
print("DBG %d %s worst %.2e %s" % (seed, "OK" if not bad else "BAD", worst, bad))
