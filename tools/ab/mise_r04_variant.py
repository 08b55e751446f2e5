"""A/B control for round 5's MISE subdivision skips: build rfdnet_amd/lib/variants/librfd_miser04.so = today's library with
round 4's mise.hip (commit 0bd4a9b: every round stages every slab of every proposal) plus a shim for the entry point that
did not exist then.  Dev container only (needs the git history); the .so travels with gpurun.
    python tools/ab/mise_r04_variant.py  &&  RFD_HIP_LIB=rfdnet_amd/lib/variants/librfd_miser04.so python bench.py --config mise128 ..."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rfdnet_amd import build as B  # noqa: E402

SHIM = r'''
// round-5 entry point on round 4's kernels: `evaluated` ignored (every proposal, every slab, every round)
RFD_API int rfd_mise_subdivide_active(int K, int res0, int depth, double threshold, const float *values,
                                      unsigned char *pstate, unsigned char *vstate, const int *evaluated, void *stream) {
  (void)evaluated;
  return rfd_mise_subdivide(K, res0, depth, threshold, values, pstate, vstate, stream);
}
'''
out_dir = os.path.join(B.LIB_DIR, "variants")
src_dir = os.path.join(out_dir, "src")
os.makedirs(src_dir, exist_ok=True)
old = subprocess.check_output(["git", "show", "0bd4a9b:rfdnet_amd/csrc/mise.hip"], cwd=ROOT).decode()
path = os.path.join(src_dir, "mise.hip")
open(path, "w").write(old + SHIM)
srcs = [path if os.path.basename(s) == "mise.hip" else s for s in B.sources()]
so = os.path.join(out_dir, "librfd_miser04.so")
subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + B.HIPCC_FLAGS + ["-I" + B.CSRC, "-o", so] + srcs)
print(so)
