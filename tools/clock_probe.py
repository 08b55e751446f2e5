"""Sample sclk / power (rocm-smi) while the fused decoder runs back to back for a few seconds."""
import os, subprocess, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rfdnet_amd import synthetic
from rfdnet_amd.iscnet.occ_decoder import DecoderCBatchNorm
K, T = 256, 32768
dec = DecoderCBatchNorm(dim=3, z_dim=32, c_dim=512, hidden_size=256); synthetic.load_seeded(dec, 1); dec = dec.cuda().eval()
p = (torch.rand(K, T, 3, device="cuda") - 0.5) * 1.1
samples = []
stop = False
def sampler():
    while not stop:
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True)
        samples.append((time.time(), r.stdout.strip()[:600]))
        time.sleep(0.2)
with torch.no_grad():
    table, fcp = dec.fold(torch.zeros(K, 32, device="cuda"), torch.randn(K, 512, device="cuda"))
    tile_prop = torch.arange(K, dtype=torch.int32, device="cuda").repeat_interleave(T // 128)
    pts = p.reshape(-1, 3).contiguous()
    dec.decode_tiles(pts, tile_prop, table, fcp); torch.cuda.synchronize()
    print("idle:", subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True).stdout.strip()[:600])
    th = threading.Thread(target=sampler); th.start()
    t0 = time.time()
    while time.time() - t0 < 4.0:
        for _ in range(10): dec.decode_tiles(pts, tile_prop, table, fcp)
        torch.cuda.synchronize()
    stop = True; th.join()
import json, re
clk, pw = [], []
for t, s_ in samples[2:]:
    m = re.search(r'sclk clock speed:": "\((\d+)Mhz', s_); w = re.search(r'Power \(W\)": "([\d.]+)', s_)
    if m and w: clk.append(int(m.group(1))); pw.append(float(w.group(1)))
print("PROBE %s kernel=%s: sclk %.0f MHz (min %d max %d), power %.0f W over %d samples" % (os.environ.get("AB_NAME", "base"), os.environ.get("RFD_DECODER_KERNEL", "w8"), sum(clk)/len(clk), min(clk), max(clk), sum(pw)/len(pw), len(clk)))
