"""Energy ledger of the fused decoder (round 4, verdict item 2): for every variant -- a side build of librfd_hip.so
(tools/ab/build_variants.py) or the four-wave kernel -- ms per launch, shader clock, socket power, JOULES per launch
(W x ms) and cycles (ms x MHz), 256 proposals x 32 768 points per launch, the kernel running back to back for 4 s
while rocm-smi is sampled every ~0.2 s.  One subprocess per variant, two interleaved repetitions.

    python tools/ab/energy_ledger.py [out.txt]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
VARIANTS = [  # (name, library variant or None = shipped, extra env, what)
    ("base", None, {}, "shipped build (three f16 MFMAs per product, chunk claiming)"),
    ("w4", None, {"RFD_DECODER_KERNEL": "w4"}, "four-wave kernel (32x32x16 MFMA, half the LDS fragment reads per tile)"),
    ("wlo0", "wlo0", {}, "W_lo fragments all zero (a third of the MFMAs multiply by zero)  [wrong results]"),
    ("nodma", "nodma", {}, "no LDS-DMA weight stream after the priming (stale, non-zero fragments)  [wrong]"),
    ("noread", "noread", {}, "no LDS fragment reads: weight operands = zeros, LDS-DMA still runs  [wrong]"),
    ("neither", "neither", {}, "no reads, no DMA, zero weight operands = pure issue time  [wrong]"),
    ("thin", "thin", {}, "NEW (i): fragment reads thinned to 1/8, REAL non-zero operands  [wrong]"),
    ("junk", "junk", {}, "control of (ii): corrections stay f16 MFMAs but accumulate apart from the data path  [wrong]"),
    ("bf16c1", "bf16c1", {}, "NEW (ii-a): W_lo x a_hi correction as a bf16 MFMA on the same bits  [wrong]"),
    ("bf16c2", "bf16c2", {}, "NEW (ii): both corrections as bf16 MFMAs on the same bits  [wrong]"),
]
FLAGS = {"wlo0": "-DDEC8_WLO_BITS=0", "nodma": "-DDEC8_NODMA=1", "noread": "-DDEC8_NOREAD=1",
         "neither": "-DDEC8_NODMA=1 -DDEC8_NOREAD=1", "thin": "-DDEC8_THIN=1", "junk": "-DDEC8_JUNK=1",
         "bf16c1": "-DDEC8_BF16C=1", "bf16c2": "-DDEC8_BF16C=2"}

CODE = r'''
import os, re, subprocess, sys, threading, time
import numpy as np, torch
sys.path.insert(0, %r)
from rfdnet_amd import synthetic, _lib
from rfdnet_amd.iscnet.occ_decoder import DecoderCBatchNorm
fx = np.load(os.path.join(%r, "tests", "golden", "F_DEC.npz"))
d = DecoderCBatchNorm(dim=3, z_dim=32, c_dim=512, hidden_size=256); synthetic.load_seeded(d, int(fx["seed"])); d = d.cuda().eval()
d.check_range = False
with torch.no_grad():
    o = d(torch.from_numpy(fx["p"]).cuda(), torch.from_numpy(fx["z"]).cuda(), torch.from_numpy(fx["c"]).cuda())
err = float(np.nan_to_num(np.abs(o.cpu().numpy() - fx["logits"]), nan=9e9).max())
K, T = 256, 32768
dec = DecoderCBatchNorm(dim=3, z_dim=32, c_dim=512, hidden_size=256); synthetic.load_seeded(dec, 1); dec = dec.cuda().eval()
g = torch.Generator(device="cuda").manual_seed(0)
p = (torch.rand(K, T, 3, device="cuda", generator=g) - 0.5) * 1.1
samples, stop = [], False
def smi():
    r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True).stdout
    m = re.search(r'sclk clock speed:": "\((\d+)Mhz', r); w = re.search(r'Power \(W\)": "([\d.]+)', r)
    return (int(m.group(1)), float(w.group(1))) if m and w else None
def sampler():
    while not stop:
        s = smi()
        if s: samples.append(s)
        time.sleep(0.15)
with torch.no_grad():
    table, fcp = dec.fold(torch.zeros(K, 32, device="cuda"), torch.randn(K, 512, device="cuda", generator=g))
    tile_prop = torch.arange(K, dtype=torch.int32, device="cuda").repeat_interleave(T // 128)
    pts = p.reshape(-1, 3).contiguous()
    for _ in range(3): dec.decode_tiles(pts, tile_prop, table, fcp)
    torch.cuda.synchronize()
    th = threading.Thread(target=sampler); th.start()
    time.sleep(0.3)
    n, t0 = 0, time.time()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < 4.0:
        for _ in range(8): dec.decode_tiles(pts, tile_prop, table, fcp)
        n += 8
        torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    stop = True; th.join()
ms = e0.elapsed_time(e1) / n
samples = samples[3:-1] or samples
clk = sum(s[0] for s in samples) / len(samples); pw = sum(s[1] for s in samples) / len(samples)
print("RESULT %%s ms %%.3f mhz %%.0f w %%.0f n %%d err %%.2e" %% (os.environ.get("AB_NAME"), ms, clk, pw, len(samples), err))
''' % (ROOT, ROOT)


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else None
    rows = {}
    for rep in range(2):
        for name, var, env_x, _ in VARIANTS:
            env = dict(os.environ, AB_NAME=name, **env_x)
            if var:
                env["RFD_HIP_LIB"] = os.path.join(ROOT, "rfdnet_amd", "lib", "variants", "librfd_%s.so" % var)
                if not os.path.exists(env["RFD_HIP_LIB"]):
                    print("missing variant", var)
                    continue
            r = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
            m = re.search(r"RESULT \S+ ms ([\d.]+) mhz (\d+) w (\d+) n (\d+) err (\S+)", r.stdout)
            if not m:
                print("FAILED", name, (r.stderr or r.stdout)[-400:])
                continue
            rows.setdefault(name, []).append(tuple(float(x) for x in m.groups()))
            print(name, m.group(0))
            sys.stdout.flush()
    lines = ["%-8s %9s %9s %7s %7s %9s %9s  %9s  %s" % ("variant", "ms rep1", "ms rep2", "MHz", "W", "J/launch", "Mcycles",
                                                        "err F_DEC", "what")]
    base = None
    for name, _, _, what in VARIANTS:
        if name not in rows:
            continue
        r = rows[name]
        ms = sum(x[0] for x in r) / len(r)
        mhz = sum(x[1] for x in r) / len(r)
        w = sum(x[2] for x in r) / len(r)
        if name == "base":
            base = ms
        lines.append("%-8s %9.3f %9.3f %7.0f %7.0f %9.2f %9.1f  %9.2e  %s%s"
                     % (name, r[0][0], r[-1][0], mhz, w, w * ms * 1e-3, ms * mhz * 1e-3, r[0][4], what,
                        "   (%+.1f %% vs base)" % (100 * (ms / base - 1)) if base and name != "base" else ""))
    text = "\n".join(lines)
    print(text)
    if out:
        with open(out, "w") as fh:
            fh.write(text + "\n")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--build":
        args = ["%s=%s" % (k, v) for k, v in FLAGS.items()]
        sys.exit(subprocess.call([sys.executable, os.path.join(ROOT, "tools", "ab", "build_variants.py")] + args))
    main()
