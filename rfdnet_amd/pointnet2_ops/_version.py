# the version string of the package this one stands in for (reference pointnet2_ops/_version.py:1)
__version__ = "3.0.0"
