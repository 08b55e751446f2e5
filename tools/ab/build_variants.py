"""Build side variants of librfd_hip.so for A/B timing:
  python tools/ab/build_variants.py NAME="-DFOO=1 -DBAR=2" NAME2="..."
-> rfdnet_amd/lib/variants/librfd_NAME.so  (git-ignored; travels with gpurun).

The shipped decoder source (rfdnet_amd/csrc/occ_decoder8.hip) carries no side-build switch.  The timing-only,
wrong-result variants of rounds 2-4 (DEC8_NOREAD, DEC8_NODMA, DEC8_THIN, DEC8_JUNK, DEC8_BF16C, DEC8_WLO_BITS,
DEC8_PRIO, DEC8_ROT=0, DEC8_FENCE, DEC8_SB=0, DEC8_DMA_AUX, DEC8_NO_PROLOGUE_BARRIER) are tools/ab/dec8_ablation.patch:
every variant is built from a scratch copy of the source with that patch applied (patched_decoder_source()).
tests/test_isa_audit.py checks that the patch applies and that the patched source without switches assembles to the
shipped kernel."""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rfdnet_amd import build as B  # noqa: E402

PATCH = os.path.join(ROOT, "tools", "ab", "dec8_ablation.patch")
DEC8 = os.path.join(B.CSRC, "occ_decoder8.hip")


def patched_decoder_source(out_dir):
    """-> path of a copy of occ_decoder8.hip with the ablation switches patched back in (compile with -I csrc)"""
    os.makedirs(out_dir, exist_ok=True)
    dst = os.path.join(out_dir, "occ_decoder8.hip")
    shutil.copyfile(DEC8, dst)
    subprocess.check_call(["patch", "--quiet", "--no-backup-if-mismatch", "-p0", dst, PATCH])
    return dst


def variant_sources(out_dir):
    return [patched_decoder_source(out_dir) if os.path.samefile(s, DEC8) else s for s in B.sources()]


def main(argv):
    out_dir = os.path.join(B.LIB_DIR, "variants")
    srcs = variant_sources(os.path.join(out_dir, "src"))
    procs = []
    for arg in argv:
        name, flags = arg.split("=", 1)
        path = os.path.join(out_dir, "librfd_%s.so" % name)
        cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + B.HIPCC_FLAGS + ["-I" + B.CSRC] + flags.split() + \
            ["-o", path] + srcs
        procs.append((name, subprocess.Popen(cmd)))
    for name, p in procs:
        rc = p.wait()
        print(name, "rc", rc)
        if rc:
            sys.exit(rc)


if __name__ == "__main__":
    main(sys.argv[1:])
