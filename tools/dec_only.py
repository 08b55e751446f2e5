"""Run only the fused decoder (for PMC counter collection with rocprofv3)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rfdnet_amd import synthetic  # noqa: E402
from rfdnet_amd.iscnet.occ_decoder import DecoderCBatchNorm  # noqa: E402

mode = int(sys.argv[1]) if len(sys.argv) > 1 else 3
K, T = 256, int(sys.argv[2]) if len(sys.argv) > 2 else 32768
dec = DecoderCBatchNorm(dim=3, z_dim=32, c_dim=512, hidden_size=256)
synthetic.load_seeded(dec, 1)
dec = dec.cuda().eval()
p = (torch.rand(K, T, 3, device="cuda") - 0.5) * 1.1
with torch.no_grad():
    table, fcp = dec.fold(torch.zeros(K, 32, device="cuda"), torch.randn(K, 512, device="cuda"))
    tile_prop = torch.arange(K, dtype=torch.int32, device="cuda").repeat_interleave(T // 128)
    pts = p.reshape(-1, 3).contiguous()
    for _ in range(3):
        dec.decode_tiles(pts, tile_prop, table, fcp, mode=mode)
torch.cuda.synchronize()
