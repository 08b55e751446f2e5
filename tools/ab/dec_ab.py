"""A/B timing of decoder variants inside ONE process launch each (same box):
   python tools/ab/dec_ab.py base v1 v2 ...   (names of rfdnet_amd/lib/variants/librfd_<name>.so; 'base' = the shipped library)
Each variant runs in its own subprocess (RFD_HIP_LIB), interleaved twice to average out drift; checks the
reference fixture before timing."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CODE = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, %r)
from rfdnet_amd import synthetic, _lib
from rfdnet_amd.iscnet.occ_decoder import DecoderCBatchNorm
fx = np.load(os.path.join(%r, "tests", "golden", "F_DEC.npz"))
d = DecoderCBatchNorm(dim=3, z_dim=32, c_dim=512, hidden_size=256); synthetic.load_seeded(d, int(fx["seed"])); d = d.cuda().eval()
with torch.no_grad():
    o = d(torch.from_numpy(fx["p"]).cuda(), torch.from_numpy(fx["z"]).cuda(), torch.from_numpy(fx["c"]).cuda())
err = float(np.abs(o.cpu().numpy() - fx["logits"]).max())
K, T = 256, 32768
dec = DecoderCBatchNorm(dim=3, z_dim=32, c_dim=512, hidden_size=256); synthetic.load_seeded(dec, 1); dec = dec.cuda().eval()
p = (torch.rand(K, T, 3, device="cuda") - 0.5) * 1.1
with torch.no_grad():
    table, fcp = dec.fold(torch.zeros(K, 32, device="cuda"), torch.randn(K, 512, device="cuda"))
    tile_prop = torch.arange(K, dtype=torch.int32, device="cuda").repeat_interleave(T // 128)
    pts = p.reshape(-1, 3).contiguous()
    for _ in range(3): dec.decode_tiles(pts, tile_prop, table, fcp)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(8): dec.decode_tiles(pts, tile_prop, table, fcp)
    e1.record(); torch.cuda.synchronize()
_lib.device_status()
ms = e0.elapsed_time(e1) / 8
print("RESULT %%s %%.3f ms %%.1f TF err %%.2e" %% (os.environ.get("AB_NAME"), ms, K * T * 1312768 / ms / 1e9, err))
''' % (ROOT, ROOT)
names = sys.argv[1:] or ["base"]
for rep in range(2):
    for n in names:
        env = dict(os.environ, AB_NAME=n)
        if n != "base":
            env["RFD_HIP_LIB"] = os.path.join(ROOT, "rfdnet_amd", "lib", "variants", "librfd_%s.so" % n)
        r = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
        print(line[0] if line else "FAILED %s: %s" % (n, (r.stderr or r.stdout)[-600:]))
        sys.stdout.flush()
