"""Rebuild the round-2 FAILING decoder build and its bisect variants as side libraries (dev container: needs the git
history).  Source = occ_decoder8.hip as of commit fd00186 (scatter epilogue compiled in, no scheduling fence) with the
static priority of commit 0872592 put back:
  fdprio      s_setprio 1 for waves 4-7                      -> wrong 16-point groups in 10/10 cold processes (round 3)
  fdprio_sb   + sched_barrier(0) in front of the slab-end wait + barrier (the slab's last MFMAs cannot sink below it)
  fdprio_sb2  + the same in front of the block-input loop's barrier
-> rfdnet_amd/lib/variants/librfd_<name>.so (git-ignored, travels with gpurun); tools/ab/prio_check.py runs them."""
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
out_dir = os.path.join(B.LIB_DIR, "variants")
os.makedirs(out_dir, exist_ok=True)
tmp = os.path.join(out_dir, "_hist")
os.makedirs(tmp, exist_ok=True)
others = [s for s in B.sources() if not s.endswith("occ_decoder8.hip")]
for name, text in (("fdprio", src), ("fdprio_sb", sb), ("fdprio_sb2", sb2)):
    path = os.path.join(tmp, "occ_decoder8_%s.hip" % name)
    open(path, "w").write(text)
    cmd = ["/opt/rocm/bin/hipcc"] + B.HIPCC_FLAGS + ["-o", os.path.join(out_dir, "librfd_%s.so" % name)] + others + [path]
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    print(name, "built")
