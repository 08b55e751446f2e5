"""fp32-class linear layers on the f16 matrix cores (csrc/gemm_f16x3.hip).

`linear(x, weight, ...)` computes act(x) @ weight.T + bias (+ per-group bias)
(+ residual) for a row-major fp32 x of shape (M, K) (or a strided row view of a
wider matrix) with three f16 MFMAs per product on (hi, lo) splits of both
operands -- ~2^-20 relative error per product, the same scheme as the occupancy
decoder.  Weights are split / re-laid once and cached per parameter version.
Shapes that do not tile (M % 128, N % 128, K % 32) fall back to nothing: the
caller decides (see usable())."""
import torch

from . import _lib, occ_fold

SA = 4            # activations scaled by 2^SA before the f16 split (|a| < 4094); lower_scale() -> SA_FALLBACK
SA_FALLBACK = 1   # after an f16-range flag (status bit 4): |a| < 32752
_cache = {}


def usable(M, N, K, x):
    return (x.is_cuda and x.dtype == torch.float32 and M % 128 == 0 and N % 128 == 0 and K % 32 == 0
            and x.stride(-1) == 1 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0)


def _packed(weight):
    """weight: (N,K) fp32 tensor (may be a column slice / cat built by the caller)."""
    key = (weight.data_ptr(), weight._version, tuple(weight.shape), str(weight.device))
    hit = _cache.get(key)
    if hit is None:
        with _lib.BUILD_LOCK:
            hit = _cache.get(key)
            if hit is None:
                w = weight.detach().contiguous()
                N, K = w.shape
                sw = occ_fold.choose_kw([w])
                buf = torch.empty(_lib.lib().rfd_gemm_packed_bytes(N, K), dtype=torch.uint8, device=w.device)
                with torch.cuda.device(w.device):
                    rc = _lib.lib().rfd_gemm_pack_w(N, K, sw, w.data_ptr(), buf.data_ptr(), _lib.current_stream())
                _lib.check(rc, "rfd_gemm_pack_w")
                hit = (buf, sw, weight)       # keep the keyed tensor alive: its address must not be reused
                if len(_cache) > 256:
                    _cache.clear()
                _lib.publish(w.device)
                _cache[key] = hit
    return hit


def linear(x, weight, bias=None, gbias=None, rows_per_group=1, residual=None, relu_in=False,
           relu_out=False, out=None, pool=None, store=True, pool_signed=False):
    """x (M,K) fp32 rows (row stride >= K allowed), weight (N,K) -> (M,N).
    pool: optional (M / rows_per_group, N) ZERO-initialised tensor that receives
    max(0, out) over the rows of every group (fused max-pool + ReLU); with store=False the
    product itself is not written (returns None).  pool_signed: the pool (initialised to -inf
    by the caller) receives the plain max instead."""
    M, K = x.shape
    N = weight.shape[0]
    assert usable(M, N, K, x)
    packed, sw, _ = _packed(weight)
    if store is False:
        assert pool is not None, "store=False only makes sense with a pool"
        out = None
    elif out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=x.device)
    ldr = residual.stride(0) if residual is not None else 0
    with torch.cuda.device(x.device):
        rc = _lib.lib().rfd_gemm_f16x3(
            M, N, K, x.data_ptr(), x.stride(0), packed.data_ptr(),
            out.data_ptr() if out is not None else None, out.stride(0) if out is not None else N,
            bias.data_ptr() if bias is not None else None,
            gbias.data_ptr() if gbias is not None else None, int(rows_per_group),
            residual.data_ptr() if residual is not None else None, ldr,
            int(relu_in), int(relu_out), SA, sw, pool.data_ptr() if pool is not None else None,
            int(pool_signed), _lib.current_stream())
    _lib.check(rc, "rfd_gemm_f16x3")
    return out


def lower_scale():
    """After status bit 4 (a stored activation * 2^SA reached the f16 limit): switch every split-precision GEMM of
    this process to the fallback scale, once.  True = run the stage again; False = already at the fallback scale."""
    global SA
    if SA <= SA_FALLBACK:
        return False
    import warnings
    warnings.warn("split-precision GEMM: an activation exceeded the f16 range at scale 2^%d; re-running at 2^%d "
                  "(kept from now on)" % (SA, SA_FALLBACK), RuntimeWarning)
    SA = SA_FALLBACK
    return True


def pool_usable(M, N, K, rows_per_group):
    return M % 256 == 0 and N % 256 == 0 and K % 128 == 0 and rows_per_group % 64 == 0
