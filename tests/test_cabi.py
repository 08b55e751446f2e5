"""CPU: the C-ABI library loads and exports every symbol include/*.h declares
(no compute calls here -- there is no GPU in the CPU tier)."""
import ctypes
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = []
    for h in sorted(glob.glob(os.path.join(ROOT, "include", "*.h"))):
        src = open(h).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        src = re.sub(r"#.*", "", src)
        for m in re.finditer(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\(([^;{}]*?)\)\s*;", src, flags=re.S):
            names.append(m.group(1))
    return sorted(set(names))


def test_library_builds_and_loads():
    from rfdnet_amd import build
    path = build.build()
    assert os.path.exists(path)
    ctypes.CDLL(path)


def test_every_declared_symbol_is_exported():
    from rfdnet_amd import _lib, build
    lib = ctypes.CDLL(build.build())
    decl = declared_symbols()
    assert len(decl) >= 17, decl
    missing = [n for n in decl if not hasattr(lib, n)]
    assert not missing, missing
    # and the ctypes binding covers the same set
    assert sorted(_lib.exported_symbols()) == decl


def test_reference_wrapper_names_present():
    """the nine *_kernel_wrapper names of the reference's host layer
    (sampling.cpp:4-13, ball_query.cpp:4-6, group_points.cpp:4-10, interpolate.cpp:4-12)"""
    from rfdnet_amd import build
    lib = ctypes.CDLL(build.build())
    for n in ["gather_points_kernel_wrapper", "gather_points_grad_kernel_wrapper",
              "furthest_point_sampling_kernel_wrapper", "query_ball_point_kernel_wrapper",
              "group_points_kernel_wrapper", "group_points_grad_kernel_wrapper",
              "three_nn_kernel_wrapper", "three_interpolate_kernel_wrapper",
              "three_interpolate_grad_kernel_wrapper"]:
        assert hasattr(lib, n), n


def test_ext_module_has_the_nine_bindings():
    """bindings.cpp:6-19"""
    from rfdnet_amd.pointnet2_ops import _ext
    for n in ["gather_points", "gather_points_grad", "furthest_point_sampling", "three_nn",
              "three_interpolate", "three_interpolate_grad", "ball_query", "group_points",
              "group_points_grad"]:
        assert callable(getattr(_ext, n)), n


def test_ext_rejects_cpu_and_bad_dtypes():
    import pytest
    import torch
    from rfdnet_amd.pointnet2_ops import _ext
    x = torch.zeros(1, 8, 3)
    with pytest.raises(RuntimeError, match="CPU not supported"):
        _ext.furthest_point_sampling(x, 4)
    with pytest.raises(RuntimeError, match="must be a float tensor"):
        _ext.furthest_point_sampling(x.double(), 4)
    with pytest.raises(RuntimeError, match="must be a contiguous tensor"):
        _ext.furthest_point_sampling(torch.zeros(1, 3, 8).transpose(1, 2), 4)
    with pytest.raises(RuntimeError, match="must be an int tensor"):
        _ext.gather_points(torch.zeros(1, 3, 8), torch.zeros(1, 4, dtype=torch.int64))
