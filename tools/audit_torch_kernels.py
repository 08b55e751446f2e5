"""Which kernels of the HOST FRAMEWORK contain the packed-fp32 op_sel forms that are wrong beside another wave's MFMA on gfx950
(profiles/r06_pk_f32_hazard.txt), and are any of them on this library's path?  CPU only, ~2 minutes on 8 cores.

libtorch_hip.so carries its device code as compressed clang offload bundles ("CCOB" blobs).  Every blob is carved out, its gfx950
code object unbundled (clang-offload-bundler), disassembled (llvm-objdump) and scanned per kernel symbol for v_pk_{fma,mul,add}_f32
with an op_sel bit set.  With --names FILE (output of tools/kernel_names.py on a rocprofv3 --kernel-trace database of bench.py) the
kernels actually launched are looked up in the result.

    python tools/audit_torch_kernels.py [--names gpurun_out/r6names/headline_names.txt ...] [--lib path/to/libtorch_hip.so]
"""
import mmap
import os
import re
import struct
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

LLVM = "/opt/rocm/lib/llvm/bin"
FORM = re.compile(r"\bv_pk_(?:fma|mul|add)_f32\b")
OPSEL = re.compile(r"op_sel:\[[01,]*1")


def blobs(path):
    f = open(path, "rb")
    mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
    pos = 0
    while True:
        i = mm.find(b"CCOB", pos)
        if i < 0:
            return
        ver, method = struct.unpack_from("<HH", mm, i + 4)
        if ver == 2 and method in (0, 1, 2):
            (fsz,) = struct.unpack_from("<I", mm, i + 8)
            yield mm[i:i + fsz]
        pos = i + 4


def scan_blob(args):
    k, data, tmp = args
    b, o = os.path.join(tmp, "b%d.bin" % k), os.path.join(tmp, "c%d.o" % k)
    open(b, "wb").write(data)
    r = subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + b,
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + o], capture_output=True)
    os.remove(b)
    out = {}
    if r.returncode == 0 and os.path.exists(o) and os.path.getsize(o) > 0:
        p = subprocess.Popen([LLVM + "/llvm-objdump", "-d", "--mcpu=gfx950", o], stdout=subprocess.PIPE, text=True)
        sym = None
        for line in p.stdout:
            if line.endswith(">:\n"):
                sym = line.split("<", 1)[1][:-3]
            elif FORM.search(line):
                t = out.setdefault(sym, [0, 0])
                t[0] += 1
                t[1] += OPSEL.search(line) is not None
        p.wait()
    if os.path.exists(o):
        os.remove(o)
    return out


def main():
    lib = None
    names = []
    a = sys.argv[1:]
    while a:
        x = a.pop(0)
        if x == "--lib":
            lib = a.pop(0)
        elif x == "--names":
            names.append(a.pop(0))
    if lib is None:
        import torch
        lib = os.path.join(os.path.dirname(torch.__file__), "lib", "libtorch_hip.so")
    res = {}
    with tempfile.TemporaryDirectory() as tmp, ThreadPoolExecutor(6) as ex:
        n = 0
        for part in ex.map(scan_blob, ((k, d, tmp) for k, d in enumerate(blobs(lib)))):
            n += 1
            for s, t in part.items():
                r = res.setdefault(s, [0, 0])
                r[0] = max(r[0], t[0])
                r[1] = max(r[1], t[1])
    bad = {s: t for s, t in res.items() if t[1]}
    print("%s: %d offload bundles, %d gfx950 kernels with packed fp32, %d of them with an op_sel bit set" % (lib, n, len(res), len(bad)))
    for s, t in sorted(bad.items(), key=lambda kv: -kv[1][1])[:12]:
        print("  %4d of %4d  %s" % (t[1], t[0], s[:150]))
    for f in names:
        hit = total = 0
        print("launched kernels of %s:" % f)
        for line in open(f):
            cnt, name = line.strip().split(None, 1)
            name = name.replace(".kd", "")
            total += 1
            if name in bad:
                hit += 1
                print("  AFFECTED  %s launches  %s" % (cnt, name[:150]))
        print("  %d distinct kernels launched, %d found in the framework's scan with packed fp32 at all, %d with the op_sel form" % (
            total, sum(1 for line in open(f) if line.strip().split(None, 1)[1].replace(".kd", "") in res), hit))


if __name__ == "__main__":
    main()
