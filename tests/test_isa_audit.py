"""CPU (no GPU): static audit of the row-owner GEMM's hand-issued activation loads.

gemm_rows8_kernel loads its activations with inline-asm `global_load_dwordx4` (so that the
compiler does not drain the LDS-DMA pipeline with its own waits) and orders them with explicit
`s_waitcnt vmcnt(N)`.  The compiler cannot check that discipline; tools/audit_vmcnt.py does, on the
generated gfx950 assembly: no VGPR written by an in-flight load may be touched before a wait has
retired the load.  (The same audit flags ~190 sites in the fused ResnetBlockFC kernel that was
withdrawn after an intermittent wrong result -- see DESIGN.md 3.4b.)"""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"),
                    reason="hipcc not available")
def test_row_owner_gemm_has_no_use_before_landed(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    asm = str(tmp_path / "gemm.s")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                    "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "rfdnet_amd", "csrc"),
                    "-S", "--cuda-device-only", "-o", asm,
                    os.path.join(ROOT, "rfdnet_amd", "csrc", "gemm_f16x3.hip")],
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=str(tmp_path))
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import audit_vmcnt
    for variant in ("ILb1ELb0E", "ILb1ELb1E", "ILb0ELb0E", "ILb0ELb1E"):
        seen, problems = audit_vmcnt.audit(asm, "gemm_rows8_kernel" + variant)
        assert seen['loads'] >= 40 and seen['waits'] >= 8, seen      # the audit saw the hand-issued loads
        assert problems == [], problems[:5]
