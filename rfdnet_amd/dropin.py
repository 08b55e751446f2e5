"""Make the reference's imports of its point-op package resolve to the MI355X operator library.

    import rfdnet_amd.dropin; rfdnet_amd.dropin.install()      # before importing the reference

The repository ships both import identities as real packages (`pointnet2_ops/` and the
`external/pointnet2_ops_lib/pointnet2_ops/` overlay); `pip install -e .` or this repository on
`sys.path` is all a host application needs.  `install()` is the in-process form: it puts the repository root on
`sys.path` (ahead of the reference checkout when `overlay=True`, so that the fused module classes are used; behind
everything else otherwise, so that the reference's own Python runs on top of `pointnet2_ops._ext`), imports the
package and returns `_ext`.  See INTEGRATION.md §2."""
import importlib
import os
import sys

REPO_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def install(overlay=False):
    if overlay:
        if REPO_ROOT in sys.path:
            sys.path.remove(REPO_ROOT)
        sys.path.insert(0, REPO_ROOT)
        # namespace packages cache their search path: a portion added after `external` was first imported is only
        # seen once the cached module is dropped
        for name in [n for n in sys.modules if n == "external" or n.startswith("external.pointnet2_ops_lib")]:
            del sys.modules[name]
        importlib.invalidate_caches()
    elif REPO_ROOT not in sys.path:
        sys.path.append(REPO_ROOT)
    pkg = importlib.import_module("pointnet2_ops")
    from .pointnet2_ops import _ext
    if pkg._ext is not _ext:
        raise ImportError("another `pointnet2_ops` package (%s) is ahead of %s on sys.path"
                          % (getattr(pkg, "__file__", "?"), REPO_ROOT))
    return _ext
