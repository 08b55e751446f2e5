"""ctypes binding of librfd_hip.so -- the C-ABI boundary (include/*.h).

The product path has NO CPU fallback: if the HIP library is missing or fails to
load, importing an op raises.  (The CPU oracle under oracle/ is test
infrastructure and is never imported from this package.)
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RFD_HIP_LIB: load another build of the same library (A/B timing of kernel variants)
LIB_PATH = os.environ.get("RFD_HIP_LIB") or os.path.join(_HERE, "lib", "librfd_hip.so")

_f = C.c_void_p      # device pointers travel as integers (tensor.data_ptr())
_i = C.c_int
_fl = C.c_float

# name -> argtypes, exactly the prototypes of include/rfd_pointnet2.h / rfd_occ.h / rfd_chamfer.h
SIGNATURES = {
    "furthest_point_sampling_kernel_wrapper": [_i, _i, _i, _f, _f, _f, _f],
    "gather_points_kernel_wrapper": [_i, _i, _i, _i, _f, _f, _f, _f],
    "gather_points_grad_kernel_wrapper": [_i, _i, _i, _i, _f, _f, _f, _f],
    "query_ball_point_kernel_wrapper": [_i, _i, _i, _fl, _i, _f, _f, _f, _f],
    "group_points_kernel_wrapper": [_i, _i, _i, _i, _i, _f, _f, _f, _f],
    "group_points_grad_kernel_wrapper": [_i, _i, _i, _i, _i, _f, _f, _f, _f],
    "three_nn_kernel_wrapper": [_i, _i, _i, _f, _f, _f, _f, _f],
    "three_interpolate_kernel_wrapper": [_i, _i, _i, _i, _f, _f, _f, _f, _f],
    "three_interpolate_grad_kernel_wrapper": [_i, _i, _i, _i, _f, _f, _f, _f, _f],
    "rfd_group_concat": [_i, _i, _i, _i, _i, _fl, _i, _i, _f, _f, _f, _f, _f, _f, _f],
    "rfd_furthest_point_sampling_gather": [_i, _i, _i, _f, _f, _f, _f, _f],
    "rfd_chamfer_forward": [_i, _i, _f, _i, _f, _f, _f, _f, _f, _f],
    "rfd_chamfer_backward": [_i, _i, _f, _i, _f, _f, _f, _f, _f, _f, _f, _f],
    "rfd_sa_fused": [_i, _i, _i, _i, _i, _fl, _i, _f, _f, _f, _f, _i, _i, _i, _f, _f, _f, _f, _f, _f, _f, _f],
    "rfd_occ_pack_weights": [_f, _f, C.POINTER(C.c_int), _i, _f, _f],
    "rfd_occ_decode": [_i, _f, _f, _f, _f, _f, _f, _f, _fl, _f, _i, _f],
    "rfd_occ_pack_weights_w8": [_f, _f, C.POINTER(C.c_int), _i, _f, _f],
    "rfd_occ_decode_w8": [_i, _f, _f, _f, _f, _f, _f, _f, _fl, _f, _i, _f],
    "rfd_occ_decode_scatter_w8": [_i, _f, _f, _f, _f, _f, _f, _f, _fl, _f, _f, _f, C.c_longlong, _i, _f],
    "rfd_occ_chunk_range": [_i, _i, _i, C.POINTER(C.c_int), C.POINTER(C.c_int)],
    "rfd_occ_chunk_range_capped": [_i, _i, _i, _i, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)],
    "rfd_occ_set_launch_shape": [_i, _i, _i],
    "rfd_occ_set_tail_tiles": [_i],
    "rfd_fps_set_timeout_ms": [_i],
    "rfd_fps_set_geometry": [_i],
    "rfd_test_hold_cus": [_i, _f, _i, _f],
    "rfd_fps_test_phantom_units": [_i],
    "rfd_make_grid_points": [_i, _fl, _fl, _fl, _f, _i, _f],
    "rfd_mise_init": [_i, _i, _i, _f, _f, _f],
    "rfd_mise_count": [_i, _i, _i, _f, _f, _f],
    "rfd_mise_collect": [_i, _i, _i, _f, _f, _f, _fl, _f, _f, _f],
    "rfd_mise_scatter": [_i, _i, _i, _f, _f, _f, _f, _f, _f, _f],
    "rfd_mise_subdivide": [_i, _i, _i, C.c_double, _f, _f, _f, _f],
    "rfd_mise_subdivide_active": [_i, _i, _i, C.c_double, _f, _f, _f, _f, _f],
    "rfd_mise_subdivide_dirty": [_i, _i, _i, C.c_double, _f, _f, _f, _f, C.c_longlong, _f, _f, _f, _f, _i, _f],
    "rfd_mise_to_dense": [_i, _i, _i, _f, _f, _f],
    "rfd_points_in_boxes": [_i, _i, _i, _i, _f, _f, _f, _f],
    "rfd_nms3d": [_i, _i, C.c_double, _i, _i, _f, _f, _f, _f, _f, _f],
    "rfd_gemm_pack_w": [_i, _i, _i, _f, _f, _f],
    "rfd_gemm_f16x3": [_i, _i, _i, _f, _i, _f, _f, _i, _f, _f, _i, _f, _i, _i, _i, _i, _i, _f, _i, _f],
    "rfd_mc_classify": [_i, _i, _fl, C.c_double, _f, _f, _f, _f, _f],
    "rfd_mc_emit": [_i, _i, _fl, C.c_double, _f, _f, _f, _f, _f, _f, _f, _f],
    "rfd_mc_emit_affine": [_i, _i, _fl, C.c_double, _f, _f, _f, _f, _f, _f, _f, C.c_double, C.c_double, _f],
    "rfd_mc_blocks": [_i],
    "rfd_chain_pack": [_i, _f, _f, _f, _i, _i, _i, _f, _f],
    "rfd_chain_pool": [_i, _i, _i, _i, _f, _i, _f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _f, _f],
    "rfd_occ_fold_rows": [_i, _i, _i, _f, _f, _f, _f, _f, _f, _f, _i, _f, _f],
    "rfd_rows3_rotate_z": [_i, _i, _f, _f, _f, _f],
    "rfd_rows3_affine": [_i, _i, _f, _f, _f, _f],
    "rfd_mlp_cols": [_i, _i, _i, _f, _f, _f, _f, _f, _f, _f],
    "rfd_three_interpolate_cat": [_i, _i, _i, _i, _i, _f, _f, _f, _f, _f, _f],
    "rfd_chain_pack_n": [_i, _i, _f, _f, _f, _i, _i, _i, _f, _f],
    "rfd_chain_pool_n": [_i, _i, _i, _i, _i, _f, _i, _f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _f, _f],
    "rfd_head_pack": [_f, _f, _f, _i, _i, _i, _f, _f],
    "rfd_head_scores": [_i, _i, _f, _i, _f, _f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _f, _f],
    "rfd_pos_embed": [_i, _i, _i, _f, _i, _f, _f, _i, _f, _f, _i, _f, _i, _i, _f],
    "rfd_pos_embed_frag": [_i, _i, _i, _f, _i, _f, _f, _i, _f, _f, _i, _f, C.c_long, _i, _f],
    "rfd_rows_to_frag": [_i, _i, _f, _i, _i, _i, _f, C.c_long, _f],
    "rfd_frag_to_rows": [_i, _i, _f, C.c_long, _i, _f, _i, _f],
    "rfd_gemm_f16x3_frag": [_i, _i, _i, _f, C.c_long, _f, _f, C.c_long, _f, _f, _i, _i, _i, _i, _f, _i, _f],
}
_RESTYPES = {
    "rfd_last_error_string": C.c_char_p,
    "rfd_build_arch": C.c_char_p,
    "rfd_device_status": C.c_int,
    "rfd_occ_packed_bytes": C.c_size_t,
}
_INT_FNS = {"rfd_stream_status": [_f], "rfd_release_stream": [_f], "rfd_stream_status_snapshot": [_f, _f]}
_SIZE_FNS = {"rfd_mise_vstate_elems": [_i, _i], "rfd_mise_dirty_elems": [_i, _i], "rfd_gemm_packed_bytes": [_i, _i], "rfd_frag_bytes": [_i, _i], "rfd_chain_packed_bytes": [], "rfd_chain_packed_bytes_n": [_i], "rfd_head_packed_bytes": []}

_lib = None

# Lazily built, cached artefacts (packed weight streams, folded BatchNorms, stacked weights) are shared by every host
# thread that runs the same network (bench.py: several scenes in flight on one model).  A miss is built under this lock,
# and published -- the building stream drained -- before it is stored, so a thread that finds the entry may use it on
# ITS stream at once.
import threading as _threading
BUILD_LOCK = _threading.RLock()


def publish(device=None):
    """drain the current stream of `device`: what was just built on it is complete for every other stream"""
    import torch
    if torch.cuda.is_available():
        torch.cuda.current_stream(device).synchronize()


class RfdHipError(RuntimeError):
    pass


def lib():
    """Load librfd_hip.so (once).  Raises if it is absent -- never falls back."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RfdHipError(
                "librfd_hip.so not built (%s). Run `python -m rfdnet_amd.build` "
                "or __graft_entry__.build(); there is no CPU fallback." % LIB_PATH)
        # torch bundles its own HIP runtime (torch/lib/libamdhip64.so).  It must
        # be in the process BEFORE our library is loaded so the dynamic loader
        # resolves our DT_NEEDED libamdhip64 to that same copy: two HIP
        # runtimes in one process cannot share device pointers or streams (the
        # second one reports "no ROCm-capable device").
        import torch  # noqa: F401
        l = C.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(l, name)
            fn.argtypes = argtypes
            fn.restype = C.c_int
        for name, rt in _RESTYPES.items():
            fn = getattr(l, name)
            fn.restype = rt
            fn.argtypes = []
        for name, at in _INT_FNS.items():
            fn = getattr(l, name)
            fn.restype = C.c_int
            fn.argtypes = at
        for name, at in _SIZE_FNS.items():
            fn = getattr(l, name)
            fn.restype = C.c_size_t
            fn.argtypes = at
        _lib = l
    return _lib


def exported_symbols():
    return sorted(list(SIGNATURES) + list(_RESTYPES) + list(_SIZE_FNS) + list(_INT_FNS))


def check(rc, what):
    """hipError_t -> Python exception (the reference exit(-1)s instead,
    cuda_utils.h:30-39)."""
    if rc != 0:
        msg = lib().rfd_last_error_string()
        raise RfdHipError("%s failed (hipError %d): %s" % (what, rc, (msg or b"").decode()))


def current_stream():
    import torch
    return torch.cuda.current_stream().cuda_stream


def _raise_status(st):
    if st < 0:
        raise RfdHipError("rfd_device_status failed")
    msgs = []
    if st & 1:
        msgs.append("furthest point sampling aborted: its workgroups were not resident together within the "
                    "exchange time-out (partitioned / CU-masked / oversubscribed GPU?)")
    if st & 2:
        msgs.append("occupancy decoder: activation exceeded the f16 range")
    if st & 4:
        msgs.append("split-precision GEMM: activation exceeded the f16 range")
    if msgs:
        e = RfdHipError("; ".join(msgs))
        e.status = st
        raise e
    return st


def raise_status(st):
    """Raise for the flags in `st` (as returned by *_status_bits)."""
    return _raise_status(st)


def stream_status_bits():
    """The current stream's status word (waits for that stream, clears the word), no exception for set flags."""
    st = lib().rfd_stream_status(current_stream())
    if st < 0:
        raise RfdHipError("rfd_stream_status failed")
    return st


_snap_pool = []
_snap_lock = _threading.Lock()


class StatusSnapshot(object):
    """The current stream's status word as it stood when the snapshot was taken (and reset, in stream order).
    Every snapshot has a pinned word of its own (from a small pool it goes back to at the first read()), so a second
    snapshot on the same stream cannot overwrite the first one's flags before they are read, and
    read() waits for the snapshot's own event: the caller need not have synchronised the stream."""

    def __init__(self):
        import torch
        s = torch.cuda.current_stream()
        with _snap_lock:
            buf = _snap_pool.pop() if _snap_pool else None
        if buf is None:
            buf = torch.zeros(1, dtype=torch.int32).pin_memory()
        self.buf, self.value = buf, None
        st = lib().rfd_stream_status_snapshot(s.cuda_stream, buf.data_ptr())
        if st < 0:
            raise RfdHipError("rfd_stream_status_snapshot failed")
        self.event = torch.cuda.Event()
        self.event.record(s)

    def read(self):
        if self.value is None:
            self.event.synchronize()
            self.value = int(self.buf[0]) & 0xffffffff
            with _snap_lock:
                if len(_snap_pool) < 64:
                    _snap_pool.append(self.buf)
            self.buf = None
        return self.value


def release_stream(stream=None):
    """Return the status slot of `stream` (default: the current one) to the pool; raises for flags still pending."""
    import torch
    st = lib().rfd_release_stream((stream or torch.cuda.current_stream()).cuda_stream)
    return _raise_status(st)


def device_status():
    """Synchronises the DEVICE; raises if a persistent kernel flagged a problem."""
    return _raise_status(lib().rfd_device_status())


def stream_status():
    """Waits for the current stream only, then reads the same status word."""
    return _raise_status(lib().rfd_stream_status(current_stream()))
