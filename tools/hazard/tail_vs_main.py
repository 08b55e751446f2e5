"""The tail decoder (csrc/occ_decoder_tail.hip) against the main kernel on one ragged launch: logits that differ, run-to-run determinism
(62 runs per arithmetic mode), and the two kernels' time for 256 full tiles.  RFD_HIP_LIB=<side build> python tools/hazard/tail_vs_main.py
compares another build of the library (tools/ab/r06_slp.sh: the one without -fno-slp-vectorize fails here, profiles/r06_pk_f32_hazard.txt)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
from rfdnet_amd import _lib, synthetic
from rfdnet_amd.iscnet.occ_decoder import DecoderCBatchNorm
from rfdnet_amd.iscnet import occ_decoder
dec = DecoderCBatchNorm(dim=3, z_dim=32, c_dim=512, hidden_size=256); synthetic.load_seeded(dec, 21); dec = dec.cuda().eval()
lib = _lib.lib()
for mode in (occ_decoder.MODE_F16X3, occ_decoder.MODE_F16X1):
    dec.mode = mode
    K = 9
    rng = np.random.default_rng(16)
    tiles = rng.integers(1, 90, K)
    tile_prop = torch.from_numpy(np.repeat(np.arange(K, dtype=np.int32), tiles)).cuda()
    n_tiles = tile_prop.shape[0]
    g = torch.Generator(device="cuda").manual_seed(16)
    pts = ((torch.rand(n_tiles * 128, 3, device="cuda", generator=g) - 0.5) * 1.1).contiguous()
    c = torch.randn(K, 512, device="cuda", generator=g)
    with torch.no_grad():
        table, fcp = dec.fold(torch.zeros(K, 32, device="cuda"), c)
        lib.rfd_occ_set_tail_tiles(0)
        ref = dec.decode_tiles(pts, tile_prop, table, fcp)
        lib.rfd_occ_set_tail_tiles(100000)
        got = dec.decode_tiles(pts, tile_prop, table, fcp)
        got2 = dec.decode_tiles(pts, tile_prop, table, fcp)
        nbad = 0
        for _ in range(30):
            nbad += int((dec.decode_tiles(pts, tile_prop, table, fcp) != ref).sum())
        print("30 more runs: differing logits in total", nbad)
        nd = 0
        for _ in range(30):
            nd += int((dec.decode_tiles(pts, tile_prop, table, fcp) != got).sum())
        print("determinism: 30 runs against the first tail run, differing logits in total", nd)
    torch.cuda.synchronize()
    d = (got - ref).abs()
    bad = d > 0
    print("mode", mode, "tiles", n_tiles, "differ", int(bad.sum()), "of", d.numel(), "max", float(d.max()), "run-to-run equal", torch.equal(got, got2))
    idx = bad.nonzero().flatten()[:40].cpu().numpy()
    print(idx, (idx % 128) // 16, idx % 16)
    print(np.bincount((bad.nonzero().flatten().cpu().numpy() % 128)//16, minlength=8))
# timing: 256 tiles, all real, logits mode
dec.mode = occ_decoder.MODE_F16X3
tile_prop = torch.arange(256, dtype=torch.int32, device="cuda")
pts = ((torch.rand(256 * 128, 3, device="cuda") - 0.5) * 1.1).contiguous()
c = torch.randn(256, 512, device="cuda")
with torch.no_grad():
    table, fcp = dec.fold(torch.zeros(256, 32, device="cuda"), c)
    for tail in (0, 100000):
        lib.rfd_occ_set_tail_tiles(tail)
        for _ in range(3):
            dec.decode_tiles(pts, tile_prop, table, fcp)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            dec.decode_tiles(pts, tile_prop, table, fcp)
        b.record(); torch.cuda.synchronize()
        print("tail", tail, "256 full tiles: %.3f ms" % (a.elapsed_time(b) / 20))
