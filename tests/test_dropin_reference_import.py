"""CPU, dev container only (skipped where /root/reference is absent, e.g. on the GPU box): the drop-in
boundary with the REFERENCE's own Python.  `rfdnet_amd.dropin.install()` makes
`import pointnet2_ops._ext` (external/pointnet2_ops_lib/pointnet2_ops/pointnet2_utils.py:8) resolve to the
MI355X operator module; the reference's autograd Functions then bind to it unchanged, and a CPU tensor
gets the reference's own error ("CPU not supported", sampling.cpp:34)."""
import importlib
import os
import sys

import pytest
import torch

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")


def test_reference_pointnet2_utils_binds_to_our_ext():
    from rfdnet_amd import dropin
    from rfdnet_amd.pointnet2_ops import _ext as ours
    saved = {k: sys.modules.get(k) for k in ("pointnet2_ops", "pointnet2_ops._ext")}
    path = os.path.join(REF, "external", "pointnet2_ops_lib", "pointnet2_ops", "pointnet2_utils.py")
    try:
        assert dropin.install() is ours
        spec = importlib.util.spec_from_file_location("ref_pointnet2_utils", path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)                      # runs `import pointnet2_ops._ext as _ext`
        assert mod._ext is ours
        for name in ("furthest_point_sample", "gather_operation", "three_nn", "three_interpolate",
                     "grouping_operation", "ball_query"):
            assert callable(getattr(mod, name))
        with pytest.raises(RuntimeError, match="CPU not supported"):
            mod.furthest_point_sample(torch.zeros(1, 16, 3), 4)
        with pytest.raises(RuntimeError, match="CPU not supported"):
            mod.ball_query(0.2, 4, torch.zeros(1, 16, 3), torch.zeros(1, 4, 3))
        q = mod.QueryAndGroup(0.2, 8, use_xyz=True)       # the reference's class on top of our ops
        assert q.radius == 0.2 and q.nsample == 8
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
