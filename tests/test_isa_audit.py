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
sys.path.insert(0, ROOT)
from rfdnet_amd.build import CODEGEN_FLAGS  # noqa: E402  (the library's own code-generation flags)


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"),
                    reason="hipcc not available")
def test_row_owner_gemm_has_no_use_before_landed(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    asm = str(tmp_path / "gemm.s")
    subprocess.run([hipcc] + CODEGEN_FLAGS + [
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
    # round 6: the persistent frag-rows kernel issues its B fragments AND its accumulator start values with opaque
    # loads; its first build let the compiler lift a matrix instruction above the `s_waitcnt vmcnt(12)` that covers its
    # operand (two sites, found by this audit, never seen to fail on hardware) -- fixed by tying the fragments to an
    # empty volatile asm behind the wait
    for variant in ("ILb1E", "ILb0E"):
        seen, problems = audit_vmcnt.audit(asm, "gemm_rowsf_kernel" + variant)
        assert seen['loads'] >= 40 and seen['waits'] >= 16, seen
        assert problems == [], problems[:5]


def _asm(tmp_path, src, name, extra=()):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    asm = str(tmp_path / name)
    path = src if os.path.isabs(src) else os.path.join(ROOT, "rfdnet_amd", "csrc", src)
    subprocess.run([hipcc] + CODEGEN_FLAGS + [
                    "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "rfdnet_amd", "csrc"),
                    "-S", "--cuda-device-only", "-o", asm, path] + list(extra),
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=str(tmp_path))
    return asm


def _ablation_source(tmp_path):
    """the decoder source with the side-build switches of rounds 2-4 patched back in (tools/ab/dec8_ablation.patch)"""
    sys.path.insert(0, os.path.join(ROOT, "tools", "ab"))
    import build_variants
    return build_variants.patched_decoder_source(str(tmp_path / "ablation_src"))


def _code_lines(asm):
    """device assembly without the lines that name the compilation unit (a hash of the source text)"""
    return [l for l in open(asm).read().splitlines()
            if "__hip_cuid_" not in l and not l.lstrip().startswith((".file", ".ident"))]


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"),
                    reason="hipcc not available")
def test_shipped_kernels_carry_no_wrong_result_switch_and_the_ablation_patch_rebuilds_them(tmp_path):
    """VERDICT round 4, hygiene: the product kernels contain no timing-only / wrong-result build switch and read no
    environment variable on the launch path; the ablation scaffolding is a patch under tools/ab/ that must keep
    applying to the shipped source, and the patched source built WITHOUT any switch is the shipped kernel,
    instruction for instruction (so a side build differs from the product by its switch alone)."""
    import re
    csrc = os.path.join(ROOT, "rfdnet_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        text = open(os.path.join(csrc, name)).read()
        assert not re.search(r"\bDEC8_[A-Z]", text), name
        assert not re.search(r"\bRFD_\w*TRACE\b", text), name          # the s_memtime stamp builds are patches too
        if name.endswith(".hip"):
            # getenv only in initialisers that run once per process (static locals / constructors)
            for m in re.finditer(r"getenv\(", text):
                line_start = text.rfind("\n", 0, m.start()) + 1
                line = text[line_start:text.find("\n", m.start())]
                assert "static const" in line or "(e = getenv" in line or "const char *to = getenv" in line, (name, line)
    shipped = _asm(tmp_path, "occ_decoder8.hip", "dec8_shipped.s")
    patched = _asm(tmp_path, _ablation_source(tmp_path), "dec8_patched.s")
    assert _code_lines(shipped) == _code_lines(patched)
    # round 6: the same rule for the frag-rows GEMM's timing-only switches (AB_NOXLOAD, AB_NODMA, ...:
    # tools/ab/gemm_frag_ablation.patch, built by tools/ab/gemm_frag_variants.py)
    assert not re.search(r"\bAB_[A-Z]", open(os.path.join(csrc, "gemm_f16x3.hip")).read())
    import build_variants
    gsrc = build_variants.patched_source("gemm_f16x3.hip", "gemm_frag_ablation.patch", str(tmp_path / "gemm_ab_src"))
    assert _code_lines(_asm(tmp_path, "gemm_f16x3.hip", "gemm_shipped.s")) == _code_lines(_asm(tmp_path, gsrc, "gemm_patched.s"))
    # the two phase-stamp patches (tools/fps_trace.py, tools/dec_trace.py) still apply and compile with their switch
    for src, patch, flag in (("sampling.hip", "fps_trace.patch", "-DRFD_FPS_TRACE"),
                             ("occ_decoder.hip", "dec4_trace.patch", "-DRFD_DECODE_TRACE")):
        traced = build_variants.patched_source(src, patch, str(tmp_path / "trace_src"))
        text = open(_asm(tmp_path, traced, src + ".trace.s", (flag,))).read()
        assert "s_memtime" in text, src
        assert "s_memtime" not in open(_asm(tmp_path, src, src + ".plain.s")).read(), src


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"),
                    reason="hipcc not available")
def test_decoder8_prefetch_never_lands_on_recently_read_mfma_sources(tmp_path):
    """The eight-wave decoder's weight-fragment prefetch (and every other LDS / global load in the kernel) must
    not write a register that one of the last SIX MFMAs read as SrcA / SrcB (unless that MFMA is 32 or more wait
    states back), nor one read as SrcC fewer than three wait states earlier (tools/audit_mfma_war.py on the shipped
    build's assembly; the round-2 build, whose prefetch reused the registers of the MFMAs issued just before, is the
    control: the audit must flag it).  No spills.  This is a conservative invariant of the generated code -- the
    structure that stays clean with a static priority at every code layout has it, round 2's does not -- and NOT the
    root cause of round 2's failure (profiles/r03_decoder_hazard.txt sections 7-9)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import audit_mfma_war
    asm = _asm(tmp_path, "occ_decoder8.hip", "dec8.s")
    text = open(asm).read()
    assert ".vgpr_spill_count: 0" in text and ".vgpr_spill_count: 1" not in text
    # "one whole k-step": six MFMAs in the three-term parity mode, two in the single-term throughput mode
    for inst, step in (("occ_decode8_kernelILi3E", 6), ("occ_decode8_kernelILi1E", 2)):
        st, problems = audit_mfma_war.audit(asm, inst, min_mfma_gap=step, min_c_states=3)
        assert st['mfma'] >= 64 and st['loads'] >= 200, st
        assert problems == [], problems[:5]
        assert st['mfma'] >= 64, st
    old = _asm(tmp_path, _ablation_source(tmp_path), "dec8_r0.s", ("-DDEC8_ROT=0", "-DDEC8_FENCE=1"))
    st, problems = audit_mfma_war.audit(old, "occ_decode8_kernelILi3E", min_mfma_gap=6, min_c_states=3)
    assert st['min_ab_gap'] == 0 and len(problems) > 50, (st, len(problems))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"),
                    reason="hipcc not available")
def test_no_kernel_runs_its_waves_at_unequal_priorities(tmp_path):
    """The one NECESSARY condition of round 2's wrong results that is understood is unequal static priorities of the
    two waves of a SIMD (the rest: a VALU move of the low-priority wave's tile prologue reading 0 for an LDS-loaded
    weight while its partner is far ahead, as a function of code layout and timing; hardware cause open --
    profiles/r03_decoder_hazard.txt sections 7-9; without `s_setprio` the failing binary itself is clean).
    Every matrix-core kernel keeps hipcc's own instruction placement somewhere, so the priority must stay out: no
    `s_setprio` anywhere in the shipped library.  (The audit's counts are printed for the record.)"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import audit_mfma_war
    report = []
    for src, kernels in (("occ_decoder.hip", ["occ_decode_kernelILi3E"]),
                         ("gemm_f16x3.hip", ["gemm_f16x3_kernel", "gemm_rows8_kernelILb1ELb0E"]),
                         ("sa_fused.hip", ["sa_fused_kernelILi4E"]),
                         ("pointseg_chain.hip", ["chain_kernelILi1E", "head_kernel"]),
                         ("occ_decoder8.hip", ["occ_decode8_kernelILi3E"])):
        asm = _asm(tmp_path, src, src + ".s")
        text = open(asm).read()
        assert "s_setprio" not in text, src
        for k in kernels:
            st, problems = audit_mfma_war.audit(asm, k, min_mfma_gap=6, min_c_states=3)
            assert st['mfma'] > 0, (src, k)
            report.append("%s %s: %d MFMAs, %d loads, closest load to an MFMA source: %s MFMAs, findings %d"
                          % (src, k, st['mfma'], st['loads'], st['min_ab_gap'], len(problems)))
    print("\n".join(report))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"),
                    reason="hipcc not available")
def test_no_two_wave_kernel_runs_a_valu_prologue_on_lds_reads_into_mfma_code_without_a_barrier(tmp_path):
    """The one code shape round 3's fault model names (profiles/r03_fault_model.txt): a long VALU prologue consuming
    LDS-read registers (fc_p of the decoder's tile: ~130 consumers, v_mov / v_pk_fma out of ds_read_b128) that runs
    into MFMA code with no s_barrier, so the SIMD partner can be in its MFMAs while this wave still reads.
    tools/audit_prologue_lds.py (control-flow-graph data flow over the generated assembly) must find NO such region in
    any kernel that runs two waves per SIMD -- round 4 found and closed two (chain_kernel<1>'s VALU first layer,
    head_kernel's 256-wide rectification) -- and must FIND the decoder's own when its barrier is compiled out (the
    positive control).  The short phase-entry stretches of the steady state are counted and printed, not refused."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import audit_prologue_lds
    report = []
    for src, kernels in (("occ_decoder8.hip", ["occ_decode8_kernelILi3E", "occ_decode8_kernelILi1E"]),
                         ("pointseg_chain.hip", ["chain_kernelILi0E", "chain_kernelILi1E", "chain_kernelILi2E", "head_kernel"]),
                         ("gemm_f16x3.hip", ["gemm_rows8_kernelILb1ELb0E", "gemm_rows8_kernelILb1ELb1E",
                                             "gemm_rows8_kernelILb0ELb0E", "gemm_rows8_kernelILb0ELb1E", "gemm_f16x3_kernel"]),
                         ("sa_fused.hip", ["sa_fused_kernelILi4E"])):
        asm = _asm(tmp_path, src, src + ".pl.s")
        for k in kernels:
            st, found = audit_prologue_lds.audit(asm, k)
            assert st['mfma'] > 0 and st['ds_reads'] > 0, (src, k, st)
            report.append("%s %s: %s; %d LDS->VALU consumers at phase entries (not refused)"
                          % (src, k, "none" if not found else "%d consumers in a prologue-class region" % len(found),
                             st['phase_entry_consumers']))
            assert found == [], (src, k, found[:3])
    control = _asm(tmp_path, _ablation_source(tmp_path), "dec8_nobar.s", ("-DDEC8_NO_PROLOGUE_BARRIER",))
    st, found = audit_prologue_lds.audit(control, "occ_decode8_kernelILi3E")
    assert len(found) >= 64, (st, len(found))
    report.append("control (decoder tile prologue without its barrier): %d consumers flagged" % len(found))
    print("\n".join(report))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"),
                    reason="hipcc not available")
def test_no_kernel_contains_packed_fp32_with_op_sel(tmp_path):
    """gfx950: v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 with an `op_sel` bit set (a source read from the HIGH half of its
    register pair for the low result) return wrong values in lanes 48-63 while another wave of the SIMD executes matrix
    instructions (tools/hazard/pk_f32_under_mfma.hip, profiles/r06_pk_f32_hazard.txt) -- round 2's wrong 16-point groups.  Any
    kernel can end up beside a matrix kernel's waves, so none may contain the form: every source of the library is compiled
    with the library's flags (rfdnet_amd/build.py: -fno-slp-vectorize keeps hipcc from forming them) and its assembly
    scanned.  The positive control: the same scan finds them in the decoder's prologue as soon as the flag is dropped."""
    import glob
    import re
    from concurrent.futures import ThreadPoolExecutor
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import audit_pk_f32
    srcs = sorted(glob.glob(os.path.join(ROOT, "rfdnet_amd", "csrc", "*.hip")))
    assert len(srcs) >= 18
    with ThreadPoolExecutor(4) as ex:
        asms = list(ex.map(lambda s: _asm(tmp_path, s, os.path.basename(s)[:-4] + ".s"), srcs))
    kernels = 0
    for a in asms:
        found, n_kernels = audit_pk_f32.audit(a)
        kernels += n_kernels
        assert found == [], (os.path.basename(a), found[:3])
    assert kernels >= 60, kernels
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    ctl = str(tmp_path / "control.s")
    subprocess.run([hipcc] + [f for f in CODEGEN_FLAGS if f != "-fno-slp-vectorize"] +
                   ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "rfdnet_amd", "csrc"), "-S",
                    "--cuda-device-only", "-o", ctl, os.path.join(ROOT, "rfdnet_amd", "csrc", "occ_decoder8.hip")],
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=str(tmp_path))
    found, _ = audit_pk_f32.audit(ctl)
    assert len(found) >= 8 and any("occ_decode8_kernel" in k for k, _, _ in found), len(found)
