"""CPU, dev container only (skipped where /root/reference is absent, e.g. on the GPU box): the drop-in boundary
exercised through the REFERENCE's own import chain, not by file path.

The reference reaches its op package under two names:
  models/iscnet/modules/pointnet2backbone.py:8    from external.pointnet2_ops_lib.pointnet2_ops.pointnet2_modules import ...
  external/pointnet2_ops_lib/pointnet2_ops/__init__.py:1-3
                                                  import pointnet2_ops.pointnet2_modules / .pointnet2_utils / ._version
  external/pointnet2_ops_lib/pointnet2_ops/pointnet2_utils.py:8
                                                  import pointnet2_ops._ext as _ext
Every case runs in a fresh interpreter (sys.path order and sys.modules are the thing under test) and must end with
the reference's modules bound to rfdnet_amd.pointnet2_ops._ext -- with `dropin.install()`, and without it (the
repository merely importable, as after `pip install -e .`)."""
import os
import subprocess
import sys
import textwrap

import pytest

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")

# bare namespace packages for `models*`, so that models/__init__.py (-> loss.py -> Chamfer JIT
# build + .cuda()) is never executed
NAMESPACE_MODELS = """
import sys, types
def ns(name, path):
    m = types.ModuleType(name); m.__path__ = [path]; sys.modules[name] = m
ns('models', REF + '/models'); ns('models.iscnet', REF + '/models/iscnet')
ns('models.iscnet.modules', REF + '/models/iscnet/modules')
"""

CHECKS = """
import torch
from rfdnet_amd.pointnet2_ops import _ext as ours
import external.pointnet2_ops_lib.pointnet2_ops as pkg
import external.pointnet2_ops_lib.pointnet2_ops.pointnet2_modules as mods
import external.pointnet2_ops_lib.pointnet2_ops.pointnet2_utils as utils
from external.pointnet2_ops_lib.pointnet2_ops.pytorch_utils import BNMomentumScheduler
import pointnet2_ops, pointnet2_ops._ext
from pointnet2_ops._version import __version__
assert pointnet2_ops._ext is ours and utils._ext is ours, (pointnet2_ops._ext, utils._ext)
assert mods.pointnet2_utils._ext is ours
assert __version__ == "3.0.0" and pkg.__version__ == "3.0.0"
origin = os.path.realpath(mods.__file__)
assert origin.startswith(EXPECT_MODS_UNDER), (origin, EXPECT_MODS_UNDER)
for name in ("furthest_point_sample", "gather_operation", "three_nn", "three_interpolate",
             "grouping_operation", "ball_query", "QueryAndGroup", "GroupAll"):
    assert callable(getattr(utils, name)), name
for name in ("PointnetSAModuleVotes", "PointnetFPModule", "STN_Group", "STN3d", "build_shared_mlp"):
    assert callable(getattr(mods, name)), name
# a CPU tensor gets the reference's own error through the reference's own Function (sampling.cpp:34)
for call in (lambda: utils.furthest_point_sample(torch.zeros(1, 16, 3), 4),
             lambda: utils.ball_query(0.2, 4, torch.zeros(1, 16, 3), torch.zeros(1, 4, 3))):
    try:
        call()
    except RuntimeError as e:
        assert "CPU not supported" in str(e), e
    else:
        raise AssertionError("no error for a CPU tensor")
# the reference's network modules, through their own import lines
import models.iscnet.modules.pointnet2backbone as bb        # :8  ...pointnet2_modules import PointnetSAModuleVotes
import models.iscnet.modules.proposal_module as pm          # :10-11
import models.iscnet.modules.skip_propagation as sp         # :9
assert bb.PointnetSAModuleVotes is mods.PointnetSAModuleVotes and bb.PointnetFPModule is mods.PointnetFPModule
assert pm.pointnet2_utils is utils and sp.STN_Group is mods.STN_Group
class Cfg:                                                  # what Pointnet2Backbone.__init__ reads (pointnet2backbone.py:21-25)
    config = {"data": {"use_color_detection": False, "no_height": False}}
net = bb.Pointnet2Backbone(Cfg())
assert sum(p.numel() for p in net.parameters()) == 641920   # SURVEY.md §8(c)
assert type(net.sa1) is mods.PointnetSAModuleVotes
try:
    net(torch.zeros(1, 4096, 4))
except RuntimeError as e:
    assert "CPU not supported" in str(e), e
else:
    raise AssertionError("backbone ran on CPU tensors")
bn = torch.nn.BatchNorm1d(4)
BNMomentumScheduler(None, torch.nn.Sequential(bn), lambda e: 0.5 * 0.5 ** e).step(2)
assert bn.momentum == 0.125
print("BOUND", origin)
"""


def run(body, path):
    """a fresh interpreter whose sys.path is exactly `path` + the standard library / site-packages"""
    env = dict(os.environ, PYTHONPATH=os.pathsep.join(path))
    code = "import os, sys\nREF = %r\nROOT = %r\n" % (REF, ROOT) + textwrap.dedent(body)
    r = subprocess.run([sys.executable, "-c", code], env=env, cwd="/tmp", capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout


REF_OPS = os.path.join(REF, "external", "pointnet2_ops_lib", "pointnet2_ops")
OUR_OPS = os.path.join(ROOT, "rfdnet_amd", "pointnet2_ops")


def test_reference_python_on_our_ext_without_install():
    """reference checkout AHEAD of this repository: the reference's own pointnet2_utils / pointnet2_modules run,
    on top of our `pointnet2_ops._ext`; nothing is installed or patched"""
    out = run(NAMESPACE_MODELS + "EXPECT_MODS_UNDER = %r\n" % REF_OPS + CHECKS, [REF, ROOT])
    assert "BOUND " + REF_OPS in out


def test_overlay_without_install():
    """this repository AHEAD of the reference: `external.pointnet2_ops_lib.pointnet2_ops` is our overlay (fused module
    classes), the rest of the reference's `external` namespace still resolves to the reference"""
    body = NAMESPACE_MODELS + "EXPECT_MODS_UNDER = %r\n" % OUR_OPS + CHECKS + """
import importlib.util
spec = importlib.util.find_spec("external.binvox_rw")           # a reference module beside the op package
assert spec is not None and spec.origin.startswith(REF), spec
"""
    out = run(body, [ROOT, REF])
    assert "BOUND " + OUR_OPS in out


def test_install_default_keeps_reference_python():
    """`dropin.install()` with only the reference on the path: appends this repository, reference Python on our _ext"""
    body = ("sys.path.insert(0, ROOT)\nfrom rfdnet_amd import dropin\nsys.path.remove(ROOT)\n"
            "assert dropin.install() is sys.modules['rfdnet_amd.pointnet2_ops._ext']\n"
            "assert sys.path[-1] == ROOT\n"
            + NAMESPACE_MODELS + "EXPECT_MODS_UNDER = %r\n" % REF_OPS + CHECKS)
    run(body, [REF])


def test_install_overlay_after_reference_namespace_was_imported():
    """`dropin.install(overlay=True)` also works late: after something already imported the reference's `external`
    namespace, the overlay still takes over `external.pointnet2_ops_lib.pointnet2_ops`"""
    body = ("import external.pointnet2_ops_lib as first          # namespace portions are cached from here on\n"
            "assert list(first.__path__) == [REF + '/external/pointnet2_ops_lib'], first.__path__\n"
            "sys.path.append(ROOT)\nfrom rfdnet_amd import dropin\n"
            "dropin.install(overlay=True)\nassert sys.path[0] == ROOT\n"
            + NAMESPACE_MODELS + "EXPECT_MODS_UNDER = %r\n" % OUR_OPS + CHECKS)
    run(body, [REF])


def test_foreign_pointnet2_ops_is_reported(tmp_path):
    """another `pointnet2_ops` ahead on the path (say, a CUDA build left in site-packages): install() says so instead
    of silently binding the reference to it"""
    other = tmp_path / "pointnet2_ops"
    other.mkdir()
    (other / "__init__.py").write_text("_ext = object()\n")
    body = ("from rfdnet_amd import dropin\n"
            "try:\n    dropin.install()\nexcept ImportError as e:\n    assert 'ahead of' in str(e), e\n"
            "else:\n    raise AssertionError('foreign package not noticed')\n")
    run(body, [str(tmp_path), ROOT])


def test_wheel_provides_both_identities(tmp_path):
    """what `pip install .` lays down (a wheel built from setup.py, unpacked -- site-packages is never touched):
    both identities importable from the unpacked tree alone, reference Python binds to its _ext"""
    import glob
    import zipfile
    r = subprocess.run([sys.executable, "-m", "pip", "wheel", ROOT, "--no-build-isolation", "--no-deps", "-q",
                        "-w", str(tmp_path / "wheel")], capture_output=True, text=True, timeout=600,
                       cwd=str(tmp_path))
    for junk in ("build", "rfdnet_amd.egg-info"):               # setuptools' in-tree droppings
        subprocess.run(["rm", "-rf", os.path.join(ROOT, junk)])
    assert r.returncode == 0, r.stdout + r.stderr
    site = tmp_path / "site"
    with zipfile.ZipFile(glob.glob(str(tmp_path / "wheel" / "rfdnet_amd-*.whl"))[0]) as z:
        names = z.namelist()
        z.extractall(site)
    assert "pointnet2_ops/__init__.py" in names
    assert "external/pointnet2_ops_lib/pointnet2_ops/__init__.py" in names
    assert "external/__init__.py" not in names and "external/pointnet2_ops_lib/__init__.py" not in names
    assert "rfdnet_amd/lib/librfd_hip.so" in names
    body = ("import pointnet2_ops._ext as e, rfdnet_amd\n"
            "assert os.path.realpath(rfdnet_amd.__file__).startswith(%r), rfdnet_amd.__file__\n"
            "import external.pointnet2_ops_lib.pointnet2_ops.pointnet2_utils as u\n"
            "assert u._ext is e and os.path.realpath(u.__file__).startswith(REF)\n" % str(site))
    run(body, [REF, str(site)])
