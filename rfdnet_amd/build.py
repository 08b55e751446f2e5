"""Build librfd_hip.so (the HIP/C-ABI library) for gfx950 with hipcc.

In-tree build: the .so lands in rfdnet_amd/lib/ so it travels with the repo
snapshot to the GPU box (it is git-ignored, not gpurun-ignored).
hipcc cross-compiles gfx950 without a GPU present.
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "librfd_hip.so")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")

# -ffp-contract=off: the fma placement in the point ops is part of the
# bit-exactness contract (csrc/common.h sumsq3), so nothing may be contracted
# implicitly.
#
# -fno-slp-vectorize: hipcc's SLP vectoriser turns pairs of fp32 operations into packed instructions (v_pk_fma_f32,
# v_pk_mul_f32, v_pk_add_f32) and feeds them through `op_sel` when an operand sits in the high half of a register
# pair.  On gfx950 such an instruction returns WRONG values in lanes 48-63 whenever another wave of the same SIMD is
# executing matrix (v_mfma) instructions -- tools/hazard/pk_f32_under_mfma.hip reproduces it in thirty lines with no
# memory access at all, profiles/r06_pk_f32_hazard.txt; it is round 2's "one channel of H' loses its y term in lanes
# 48-63" (DESIGN.md section 4).  Any kernel's waves can share a SIMD with a matrix kernel's, so NO kernel of the
# library may contain these forms: the flag removes them, tests/test_isa_audit.py checks the generated code.
CODEGEN_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize"]
HIPCC_FLAGS = CODEGEN_FLAGS + ["-fPIC", "-shared", "-fvisibility=hidden", "-I" + INCLUDE]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def is_stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(INCLUDE, "*.h")) + \
        [os.path.abspath(__file__)]                     # the flags live in this file
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not is_stale():
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(LIB_DIR, exist_ok=True)
    flags = list(HIPCC_FLAGS)
    if os.environ.get("RFD_NO_TEST_HOOKS") == "1":     # deployment build: no rfd_test_hold_cus / rfd_fps_test_phantom_units
        flags.append("-DRFD_NO_TEST_HOOKS")
    cmd = [hipcc] + flags + ["-o", LIB_PATH] + sources()
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
