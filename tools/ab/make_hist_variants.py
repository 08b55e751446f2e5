"""Rebuild the round-2 FAILING decoder build and its bisect variants as side libraries (dev container: needs the git
history).  Source = occ_decoder8.hip as of commit fd00186 (scatter epilogue compiled in, no scheduling fence) with the
static priority of commit 0872592 put back:
  fdprio      s_setprio 1 for waves 4-7                      -> wrong 16-point groups in 10/10 cold processes (round 3)
  fdprio_sb   + sched_barrier(0) in front of the slab-end wait + barrier (the slab's last MFMAs cannot sink below it)
  fdprio_sb2  + the same in front of the block-input loop's barrier
  fdprio_kb   fdprio + the a2' operands (SrcB) of the slab's last MFMAs kept ALLOCATED across the next slab's table loads
  fdprio_ka   ... and their weight fragments (SrcA) too: same schedule (the MFMAs still sink below the barrier), but the
              loads can no longer be given the registers those MFMAs read -- separates "register overlap" from "timing"
  fdprio_nN   fdprio + N wait states (s_nop) between the sunk MFMAs and the next slab's table loads, N = 1 .. 32:
              the distance at which the real kernel stops failing
  fdprio_vN   fdprio + N v_nop (VALU no-ops, which go through the vector issue port like the MFMAs do) at the same place
-> rfdnet_amd/lib/variants/librfd_<name>.so (git-ignored, travels with gpurun); tools/ab/prio_check.py runs them.
NOTE: these are SOURCE-level variants -- one added statement lets hipcc re-allocate and re-schedule the whole kernel (fdprio
vs fdprio_sb: 1258 differing assembly lines), so a clean variant says nothing about the site of the statement.
tools/ab/asm_variants.py edits the failing build's assembly instead (profiles/r03_decoder_hazard.txt section 7)."""
usage = "python tools/ab/make_hist_variants.py [name ...]   (no names: all)"
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rfdnet_amd import build as B  # noqa: E402

src = subprocess.check_output(["git", "-C", ROOT, "show", "fd00186:rfdnet_amd/csrc/occ_decoder8.hip"], text=True)
a = src.index('  // NO s_setprio here.')
b = src.index('  const int t_begin')
src = src[:a] + '  if (wave >= 4) __builtin_amdgcn_s_setprio(1);\n\n' + src[b:]
src = src.replace('#include "common.h"', '#include "%s/rfdnet_amd/csrc/common.h"' % ROOT)
src = src.replace('#include "../../include/rfd_occ.h"', '#include "%s/include/rfd_occ.h"' % ROOT)
slab_end = '''        phase_a();
        phase_b();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");'''
assert slab_end in src
sb = src.replace(slab_end, slab_end.replace('        asm volatile', '        __builtin_amdgcn_sched_barrier(0);\n        asm volatile'))
blk_end = '''      __syncthreads();

      for (int mb = 0; mb < 8; ++mb) {'''
assert blk_end in sb
sb2 = sb.replace(blk_end, '      __builtin_amdgcn_sched_barrier(0);\n' + blk_end)

def keep(text, frags):
    """Carry the last k-step's operands out of the slab and name them as inputs of an empty asm statement behind the
    next slab's conditioning-table loads."""
    names = ["kp0", "kp1"] + (["kp2", "kp3", "kp4", "kp5"] if frags else [])
    loop = "      for (int mb = 0; mb < 8; ++mb) {"
    assert text.count(loop) == 1
    text = text.replace(loop, "      half8 " + ", ".join(n + " = {}" for n in names) + ";\n" + loop)
    act = "act_kstep<X3>(acc_cur[0], acc_cur[1], S1, T1, 32 * mb + g4, bhi, blo, amax16);"
    assert text.count(act) == 1
    text = text.replace(act, act + "\n        asm volatile(\"\" :: " + ", ".join("\"v\"(%s)" % n for n in names)
                        + " : \"memory\");")
    tail = ("              Hs[2 * tp + 1] = mfma16(c1l, bhi, Hs[2 * tp + 1]);\n"
            "            }\n"
            "          }\n")
    assert text.count(tail) == 1
    out = "          kp0 = bhi; kp1 = blo;" + (" kp2 = n0h; kp3 = n0l; kp4 = n1h; kp5 = n1l;" if frags else "") + "\n"
    return text.replace(tail, tail + out)


def nops(text, n):
    """N wait states in front of the slab's conditioning-table loads (s_nop k = k + 1 wait states, k <= 15)."""
    act = "act_kstep<X3>(acc_cur[0], acc_cur[1], S1, T1, 32 * mb + g4, bhi, blo, amax16);"
    assert text.count(act) == 1
    parts = []
    while n > 0:
        k = min(n, 16)
        parts.append("s_nop %d" % (k - 1))
        n -= k
    return text.replace(act, 'asm volatile("' + "\\n\\t".join(parts) + '" ::: "memory");\n        ' + act)


def vnops(text, n):
    act = "act_kstep<X3>(acc_cur[0], acc_cur[1], S1, T1, 32 * mb + g4, bhi, blo, amax16);"
    assert text.count(act) == 1
    return text.replace(act, 'asm volatile("' + "\\n\\t".join(["v_nop"] * n) + '" ::: "memory");\n        ' + act)


out_dir = os.path.join(B.LIB_DIR, "variants")
os.makedirs(out_dir, exist_ok=True)
tmp = os.path.join(out_dir, "_hist")
os.makedirs(tmp, exist_ok=True)
others = [s for s in B.sources() if not s.endswith("occ_decoder8.hip")]
for name, text in (("fdprio", src), ("fdprio_sb", sb), ("fdprio_sb2", sb2), ("fdprio_kb", keep(src, False)),
                   ("fdprio_ka", keep(src, True))) + tuple(("fdprio_n%d" % n, nops(src, n)) for n in (1, 2, 4, 8, 16, 32, 128, 512)) + tuple(
                       ("fdprio_v%d" % n, vnops(src, n)) for n in (1, 4)):
    if len(sys.argv) > 1 and name not in sys.argv[1:]:
        continue
    path = os.path.join(tmp, "occ_decoder8_%s.hip" % name)
    open(path, "w").write(text)
    cmd = ["/opt/rocm/bin/hipcc"] + B.HIPCC_FLAGS + ["-o", os.path.join(out_dir, "librfd_%s.so" % name)] + others + [path]
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    print(name, "built")
