"""Build the reference's OWN CPU Chamfer-distance op (external/pyTorchChamferDistance/
chamfer_distance/chamfer_distance.cpp: nnsearch + backward loops) as oracle/_ref/cd_ref*.so,
from the source where it lies under /root/reference (never copied).  Only the .cpp is
compiled; the CUDA launchers it forward-declares stay undefined (they are only referenced by
the *_cuda entry points, which are never called).  Runs only where /root/reference exists;
used to pin oracle/rfd_oracle.c:oracle_chamfer_* and to generate tests/golden/F_CD.npz."""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/external/pyTorchChamferDistance/chamfer_distance/chamfer_distance.cpp"
OUT_DIR = os.path.join(HERE, "_ref")


def build():
    if not os.path.exists(SRC):
        return None
    from torch.utils import cpp_extension
    import torch
    os.makedirs(OUT_DIR, exist_ok=True)
    out = os.path.join(OUT_DIR, "cd_ref" + sysconfig.get_config_var("EXT_SUFFIX"))
    if os.path.exists(out) and os.path.getmtime(out) > os.path.getmtime(SRC):
        return out
    inc = ["-I" + p for p in cpp_extension.include_paths()] + ["-I" + sysconfig.get_paths()["include"]]
    libdir = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-w", "-DTORCH_EXTENSION_NAME=cd_ref",
           "-DTORCH_API_INCLUDE_EXTENSION_H", "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI)] + inc + \
          [SRC, "-o", out, "-L" + libdir, "-ltorch", "-ltorch_cpu", "-lc10", "-ltorch_python", "-Wl,-rpath," + libdir]
    subprocess.check_call(cmd)
    return out


def load():
    path = build()
    if path is None:
        return None
    import importlib.util
    import torch  # noqa: F401  (symbols of the extension)
    # lazy binding: the undefined CUDA launchers are only reached through the *_cuda entries
    flags = sys.getdlopenflags()
    sys.setdlopenflags(os.RTLD_LAZY | os.RTLD_GLOBAL)
    try:
        spec = importlib.util.spec_from_file_location("cd_ref", path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        sys.setdlopenflags(flags)
    return mod


if __name__ == "__main__":
    print(build())
