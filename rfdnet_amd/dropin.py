"""Make `import pointnet2_ops._ext` (external/pointnet2_ops_lib/pointnet2_ops/
pointnet2_utils.py:8 in the reference) resolve to the MI355X operator library.

    import rfdnet_amd.dropin; rfdnet_amd.dropin.install()      # before importing the reference

See INTEGRATION.md §2."""
import sys
import types


def install():
    from .pointnet2_ops import _ext
    pkg = sys.modules.get("pointnet2_ops")
    if pkg is None:
        pkg = types.ModuleType("pointnet2_ops")
        pkg.__path__ = []
        sys.modules["pointnet2_ops"] = pkg
    pkg._ext = _ext
    sys.modules["pointnet2_ops._ext"] = _ext
    return _ext
