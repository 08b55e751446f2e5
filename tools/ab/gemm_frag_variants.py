"""Build the timing-only ablation variants of gemm_rowsf_kernel (tools/ab/gemm_frag_ablation.patch):
every other source is compiled ONCE to an object, a variant is the patched gemm_f16x3.hip compiled with its -DAB_*
flags and linked against those.  -> rfdnet_amd/lib/variants/librfd_gf_<name>.so    (python tools/ab/gemm_frag_ab.py)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import build_variants as V  # noqa: E402
from rfdnet_amd import build as B  # noqa: E402

VARIANTS = {"gf_base": [], "gf_noxload": ["-DAB_NOXLOAD"], "gf_nodma": ["-DAB_NODMA"], "gf_nolds": ["-DAB_NOLDS"],
            "gf_nobar": ["-DAB_NOBARRIER"], "gf_noepi": ["-DAB_NOEPI"], "gf_nomem": ["-DAB_NOXLOAD", "-DAB_NODMA"],
            "gf_nostore": ["-DAB_NOSTORE"], "gf_nopool": ["-DAB_NOPOOL"], "gf_noxwait": ["-DAB_NOXWAIT"], "gf_xsame": ["-DAB_XSAME=8"],
            "gf_xsame_nostore": ["-DAB_XSAME=8", "-DAB_NOSTORE"], "gf_xmall512": ["-DAB_XSAME=512"],
            "gf_xmall1024": ["-DAB_XSAME=1024"], "gf_xmall2048": ["-DAB_XSAME=2048"],
            "gf_mfmaonly": ["-DAB_NOXLOAD", "-DAB_NODMA", "-DAB_NOLDS", "-DAB_NOBARRIER", "-DAB_NOEPI"]}


def main(names):
    out = os.path.join(B.LIB_DIR, "variants")
    obj = os.path.join(out, "obj")
    os.makedirs(obj, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cflags = [f for f in B.HIPCC_FLAGS if f != "-shared"] + ["-I" + B.CSRC, "-c"]
    others = [s for s in B.sources() if os.path.basename(s) != "gemm_f16x3.hip"]
    jobs = []
    for s in others:
        o = os.path.join(obj, os.path.basename(s) + ".o")
        if not os.path.exists(o) or os.path.getmtime(o) < os.path.getmtime(s):
            jobs.append(subprocess.Popen([hipcc] + cflags + ["-o", o, s]))
    src = V.patched_source("gemm_f16x3.hip", "gemm_frag_ablation.patch", os.path.join(out, "src"))
    vobjs = {}
    for n in names:
        vobjs[n] = os.path.join(obj, n + ".o")
        jobs.append(subprocess.Popen([hipcc] + cflags + VARIANTS[n] + ["-o", vobjs[n], src]))
        if len(jobs) >= 8:
            for j in jobs:
                assert j.wait() == 0
            jobs = []
    for j in jobs:
        assert j.wait() == 0
    for n in names:
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
                               os.path.join(out, "librfd_%s.so" % n), vobjs[n]] +
                              [os.path.join(obj, os.path.basename(s) + ".o") for s in others])
        print("built", n)


if __name__ == "__main__":
    main(sys.argv[1:] or sorted(VARIANTS))
