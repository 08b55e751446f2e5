"""Debug: build the library with -DRFD_DECODE_TRACE into a side .so and print the
s_memtime phase stamps of the decoder's block 1 (cycles, relative)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rfdnet_amd import _lib, build, synthetic  # noqa: E402

# the stamps are not in the product source: tools/ab/dec4_trace.patch puts them into a scratch copy of occ_decoder.hip
sys.path.insert(0, os.path.join(ROOT, "tools", "ab"))
import build_variants  # noqa: E402
so = os.path.join(ROOT, "rfdnet_amd", "lib", "variants", "librfd_dec4_trace.so")
if not os.path.exists(so) or "--rebuild" in sys.argv or "--build-only" in sys.argv:
    build_variants.build_patched(so, "occ_decoder.hip", "dec4_trace.patch", ["-DRFD_DECODE_TRACE"])
if "--build-only" in sys.argv:
    sys.exit(0)
_lib.LIB_PATH = so
from rfdnet_amd.iscnet.occ_decoder import DecoderCBatchNorm  # noqa: E402

mode = 3
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
        out = dec.decode_tiles(pts, tile_prop, table, fcp, mode=mode)
torch.cuda.synchronize()
raw = out[:64 * 128].cpu().numpy().view(np.uint64).reshape(64, 64)
names = ["blk_start", "conv_done", "gemm1_0_done"]
for mb in range(8):
    names += ["mb%d_start" % mb, "mb%d_phaseA" % mb, "mb%d_phaseB" % mb, "mb%d_vmcnt" % mb, "mb%d_barrier" % mb]
for tile in (0, 1, 17, 40):
    st = raw[tile][:len(names)].astype(np.int64)
    d = np.diff(st)
    print("tile", tile, "block-1 total", st[-1] - st[0])
    print("  conv %d  gemm1_0 %d  sync %d" % (d[0], d[1], st[3] - st[2]))
    for mb in range(8):
        b = 3 + 5 * mb
        print("  mb%d: phaseA %5d  phaseB %5d  vmcnt-wait %5d  barrier %5d" %
              (mb, st[b + 1] - st[b], st[b + 2] - st[b + 1], st[b + 3] - st[b + 2], st[b + 4] - st[b + 3]))
