"""Time the fused decoder alone (8.4M points, parity mode)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rfdnet_amd import synthetic, _lib

from rfdnet_amd.iscnet.occ_decoder import DecoderCBatchNorm
K, T = 256, 32768
dec = DecoderCBatchNorm(dim=3, z_dim=32, c_dim=512, hidden_size=256)
synthetic.load_seeded(dec, 1)
dec = dec.cuda().eval()
p = (torch.rand(K, T, 3, device="cuda") - 0.5) * 1.1
with torch.no_grad():
    table, fcp = dec.fold(torch.zeros(K, 32, device="cuda"), torch.randn(K, 512, device="cuda"))
    tile_prop = torch.arange(K, dtype=torch.int32, device="cuda").repeat_interleave(T // 128)
    pts = p.reshape(-1, 3).contiguous()
    for _ in range(2):
        dec.decode_tiles(pts, tile_prop, table, fcp)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        dec.decode_tiles(pts, tile_prop, table, fcp)
    e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print("decoder %.3f ms  %.1f TFLOP/s algorithmic" % (ms, K * T * 1312768 / ms / 1e9))
