"""Version of the op package this one stands in for: the reference's `pointnet2_ops/_version.py` says 3.0.0, and its
package `__init__` does `from pointnet2_ops._version import __version__` (external/pointnet2_ops_lib/pointnet2_ops/
__init__.py:3), so the name must exist under both import identities."""
VERSION_INFO = (3, 0, 0)
__version__ = ".".join(str(part) for part in VERSION_INFO)
