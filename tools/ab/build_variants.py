"""Build side variants of librfd_hip.so for A/B timing (tools/ab/run_ab.sh):
  python tools/ab/build_variants.py NAME="-DFOO=1 -DBAR=2" NAME2="..."
-> rfdnet_amd/lib/variants/librfd_NAME.so  (git-ignored; travels with gpurun)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rfdnet_amd import build as B  # noqa: E402

out_dir = os.path.join(B.LIB_DIR, "variants")
os.makedirs(out_dir, exist_ok=True)
procs = []
for arg in sys.argv[1:]:
    name, flags = arg.split("=", 1)
    path = os.path.join(out_dir, "librfd_%s.so" % name)
    cmd = ["/opt/rocm/bin/hipcc"] + B.HIPCC_FLAGS + flags.split() + ["-o", path] + B.sources()
    procs.append((name, subprocess.Popen(cmd)))
for name, p in procs:
    rc = p.wait()
    print(name, "rc", rc)
    if rc:
        sys.exit(rc)
