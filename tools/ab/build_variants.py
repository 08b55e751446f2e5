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


def patched_source(src_name, patch_name, out_dir):
    """-> path of a scratch copy of rfdnet_amd/csrc/<src_name> with tools/ab/<patch_name> applied (compile with -I csrc).
    The patches hold every build switch that makes a kernel compute something else than the product does: the decoder's
    ablation switches (dec8_ablation.patch) and the s_memtime phase stamps of tools/fps_trace.py / tools/dec_trace.py
    (fps_trace.patch, dec4_trace.patch: RFD_FPS_TRACE / RFD_DECODE_TRACE overwrite an output buffer with time stamps)."""
    os.makedirs(out_dir, exist_ok=True)
    dst = os.path.join(out_dir, src_name)
    shutil.copyfile(os.path.join(B.CSRC, src_name), dst)
    subprocess.check_call(["patch", "--quiet", "--no-backup-if-mismatch", "-p0", dst,
                           os.path.join(ROOT, "tools", "ab", patch_name)])
    return dst


def patched_decoder_source(out_dir):
    """-> path of a copy of occ_decoder8.hip with the ablation switches patched back in"""
    return patched_source("occ_decoder8.hip", "dec8_ablation.patch", out_dir)


def build_patched(so_path, src_name, patch_name, flags):
    """today's library with ONE source replaced by its patched copy, built with `flags` -> so_path"""
    src = patched_source(src_name, patch_name, os.path.join(B.LIB_DIR, "variants", "src"))
    srcs = [src if os.path.basename(s) == src_name else s for s in B.sources()]
    subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + B.HIPCC_FLAGS + ["-I" + B.CSRC] + list(flags) +
                          ["-o", so_path] + srcs)
    return so_path


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
