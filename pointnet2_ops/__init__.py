"""The pip-installed identity of the point-op package (reference external/pointnet2_ops_lib/setup.py:28 installs
`pointnet2_ops` with the extension `pointnet2_ops._ext`), provided by the MI355X library.

The reference reaches the package under two names:
  * `pointnet2_ops._ext` / `pointnet2_ops.pointnet2_utils` / `pointnet2_ops.pointnet2_modules` / `pointnet2_ops._version`
    -- `external/pointnet2_ops_lib/pointnet2_ops/__init__.py:1-3`, `pointnet2_utils.py:8`;
  * `external.pointnet2_ops_lib.pointnet2_ops.{pointnet2_utils,pointnet2_modules,pytorch_utils}`
    -- `models/iscnet/modules/pointnet2backbone.py:8`, `proposal_module.py:10-11`, `skip_propagation.py:9`,
    `net_utils/libs.py:8`, `models/optimizers.py:5` (this repository's `external/` overlay, or the reference's own files
    on top of this package's `_ext`).
Every submodule here IS the `rfdnet_amd.pointnet2_ops` module of the same name (same module object)."""
import importlib
import sys

for _name in ("_ext", "pointnet2_utils", "pointnet2_modules", "pytorch_utils", "_version"):
    _mod = importlib.import_module("rfdnet_amd.pointnet2_ops." + _name)
    sys.modules[__name__ + "." + _name] = _mod
    globals()[_name] = _mod
__version__ = _version.__version__          # noqa: F821
del importlib, sys, _name, _mod
