"""`pip install -e .` (or a wheel) provides the three import roots of the drop-in:

  rfdnet_amd                                   the library (librfd_hip.so is built in-tree first: `python -m rfdnet_amd.build`)
  pointnet2_ops                                the pip identity of the reference's op package (its setup.py:28)
  external.pointnet2_ops_lib.pointnet2_ops     overlay for the reference's in-tree identity (pointnet2backbone.py:8);
                                               `external` / `external.pointnet2_ops_lib` stay namespace packages, as in
                                               the reference's tree, so the rest of the reference's `external/` still resolves
"""
from setuptools import setup

setup(
    name="rfdnet-amd",
    version="0.6.0",
    description="MI355X-native hot path of RfD-Net behind the reference's pointnet2_ops interface",
    python_requires=">=3.10",
    packages=["rfdnet_amd", "rfdnet_amd.iscnet", "rfdnet_amd.pointnet2_ops", "pointnet2_ops",
              "external.pointnet2_ops_lib.pointnet2_ops"],
    package_data={"rfdnet_amd": ["lib/*.so", "csrc/*.hip", "csrc/*.h"]},
    install_requires=["torch", "numpy"],
    zip_safe=False,
)
