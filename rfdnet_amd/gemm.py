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


# ---- fragment-ordered split activations ("frag rows", csrc/gemm_f16x3.hip) -----------------------------------------
# A frag buffer of M rows x C channels is an f16 tensor of shape (M/32, C/32, 2, 2, 64, 8) = [row block][channel block]
# [k step][hi / lo][lane][8]: relu(x) * 2^sa split into f16 (hi, lo), in the operand order of the consumer's matrix
# instruction.  A channel window is a slice of dim 1 (row-block stride = stride(0)), so [hidden | input] concatenations
# are views.  The scale exponent `sa` is a property of the DATA: producer and consumer of a buffer must use the same one
# (callers capture gemm.SA once per forward pass).
FRAG_SHAPE = (2, 2, 64, 8)


def frag_empty(M, C, device):
    assert M % 32 == 0 and C % 32 == 0
    return torch.empty((M // 32, C // 32) + FRAG_SHAPE, dtype=torch.float16, device=device)


def _frag_args(f):
    """(data_ptr, row-block stride in bytes) of a frag tensor or a dim-1 window of one"""
    assert f.dtype == torch.float16 and f.shape[2:] == FRAG_SHAPE and f.stride()[1:] == (2048, 1024, 512, 8, 1), \
        "not a frag buffer (or not a plain channel window of one)"
    return f.data_ptr(), f.stride(0) * 2


def frag_usable(M, N, K, rows_per_group=64):
    """shapes rfd_gemm_f16x3_frag takes"""
    return M % 256 == 0 and N % 256 == 0 and K % 128 == 0 and rows_per_group % 64 == 0


def rows_to_frag(x, sa=None, relu=True, out=None):
    """x (M,C) fp32 rows (row stride allowed) -> frag buffer of relu?(x) * 2^sa"""
    M, Cc = x.shape
    sa = SA if sa is None else sa
    assert x.is_cuda and x.dtype == torch.float32 and x.stride(1) == 1
    out = frag_empty(M, Cc, x.device) if out is None else out
    ptr, stride = _frag_args(out)
    with torch.cuda.device(x.device):
        rc = _lib.lib().rfd_rows_to_frag(M, Cc, x.data_ptr(), x.stride(0), int(relu), int(sa), ptr, stride,
                                         _lib.current_stream())
    _lib.check(rc, "rfd_rows_to_frag")
    return out


def frag_to_rows(f, sa=None):
    """frag buffer (or channel window) -> (M,C) fp32 = (hi + lo) * 2^-sa (exact)"""
    sa = SA if sa is None else sa
    M, Cc = f.shape[0] * 32, f.shape[1] * 32
    x = torch.empty(M, Cc, dtype=torch.float32, device=f.device)
    ptr, stride = _frag_args(f)
    with torch.cuda.device(f.device):
        rc = _lib.lib().rfd_frag_to_rows(M, Cc, ptr, stride, int(sa), x.data_ptr(), Cc, _lib.current_stream())
    _lib.check(rc, "rfd_frag_to_rows")
    return x


def linear_frag(a, weight, bias=None, gbias=None, rows_per_group=1, out=None, pool=None, store=True,
                pool_signed=False, sa=None):
    """split(relu(A @ weight.T + bias + gbias[row // rows_per_group]) * 2^sa) with A given as a frag buffer / window
    (already rectified and scaled by ITS producer at the same sa).  out: frag buffer / window for the result (allocated
    when None and store); pool: (M / rows_per_group, N) receives the max over each group's rows of the fp32 result
    (max(0, .) into zeros; the plain max into -inf with pool_signed).  Returns out (None with store=False)."""
    M, K = a.shape[0] * 32, a.shape[1] * 32
    N = weight.shape[0]
    sa = SA if sa is None else sa
    assert gbias is None or (gbias.stride(1) == 1 and gbias.shape[1] == N), "gbias: (groups, N) rows, row stride allowed"
    assert weight.shape[1] == K and frag_usable(M, N, K, rows_per_group if (gbias is not None or pool is not None) else 64)
    packed, sw, _ = _packed(weight)
    if not store:
        assert pool is not None, "store=False only makes sense with a pool"
        out = None
    elif out is None:
        out = frag_empty(M, N, a.device)
    aptr, astride = _frag_args(a)
    cptr, cstride = _frag_args(out) if out is not None else (None, 0)
    with torch.cuda.device(a.device):
        rc = _lib.lib().rfd_gemm_f16x3_frag(
            M, N, K, aptr, astride, packed.data_ptr(), cptr, cstride,
            bias.data_ptr() if bias is not None else None, gbias.data_ptr() if gbias is not None else None,
            int(gbias.stride(0)) if gbias is not None else 0,
            int(rows_per_group), int(sa), sw, pool.data_ptr() if pool is not None else None, int(pool_signed),
            _lib.current_stream())
    _lib.check(rc, "rfd_gemm_f16x3_frag")
    return out
