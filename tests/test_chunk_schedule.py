"""CPU: the eight-wave decoder's run-time tile schedule (csrc/occ_decoder8.hip chunk_range, exported as
rfd_occ_chunk_range -- pure host arithmetic, no GPU): for every launch size and grid size the chunks partition
[0, n_tiles) in order, without gaps or overlaps; the first batch is half of the launch in equal chunks, the last
chunks are single tiles (the kernel's tail is one tile), and past the last chunk the range is empty.

The reference walks proposals and <=100 000-point slices one after the other (generator.py:71-74, 129-141); how the
tiles of ONE launch are spread over the chip has no counterpart there -- results do not depend on it
(tests/test_gpu_decoder.py compares claimed vs static launches bit for bit)."""
import ctypes as C

import pytest


def _ranges(lib, n, W):
    out, k = [], 0
    b, e = C.c_int(), C.c_int()
    while True:
        assert lib.rfd_occ_chunk_range(k, n, W, C.byref(b), C.byref(e)) == 0
        if b.value >= n:
            assert b.value == e.value == n
            break
        out.append((b.value, e.value))
        k += 1
        assert k < 10 * W + 100000
    # a few indices past the end stay empty (late workgroups' failed claims)
    for kk in (k, k + 1, k + W, k + 7 * W + 3):
        lib.rfd_occ_chunk_range(kk, n, W, C.byref(b), C.byref(e))
        assert b.value == e.value == n, (kk, b.value, e.value)
    return out


@pytest.mark.parametrize("n,W", [(1, 256), (5, 256), (255, 256), (256, 256), (257, 256), (11327, 256), (11327, 252),
                                 (71936, 256), (71936, 244), (524288, 256), (1000, 1), (1000, 3), (2 ** 24, 256)])
def test_chunks_partition_the_launch(n, W):
    from rfdnet_amd import _lib
    lib = _lib.lib()
    r = _ranges(lib, n, W)
    assert r[0][0] == 0 and r[-1][1] == n
    for (b0, e0), (b1, e1) in zip(r, r[1:]):
        assert e0 == b1 and b0 < e0
    sizes = [e - b for b, e in r]
    assert sizes == sorted(sizes, reverse=True) or n < 2 * W     # never growing (a partial batch may end short)
    assert sizes[-1] == 1 or len(r) <= W                          # tail = single tiles once there is more than one batch
    if n >= 4 * W:
        first = sizes[:W]
        assert len(set(first)) == 1 and abs(sum(first) - n / 2) <= W     # first batch: half of the launch, equal chunks
        # ~log2 chunks per workgroup: few conditioning-table loads
        assert len(r) <= W * 20


def test_bad_arguments_are_refused():
    from rfdnet_amd import _lib
    lib = _lib.lib()
    b, e = C.c_int(), C.c_int()
    assert lib.rfd_occ_chunk_range(-1, 10, 4, C.byref(b), C.byref(e)) != 0
    assert lib.rfd_occ_chunk_range(0, 10, 0, C.byref(b), C.byref(e)) != 0


@pytest.mark.parametrize("n,W,cap", [(1, 256, 16), (300, 256, 16), (11327, 256, 16), (71936, 256, 16), (71936, 256, 8),
                                     (71936, 244, 32), (524288, 256, 16), (1000, 3, 5), (5000, 256, 255)])
def test_capped_chunks_partition_the_launch_and_never_exceed_the_cap(n, W, cap):
    """the one-workgroup-per-chunk launch (RFD_DECODER_CHUNK): same schedule, no chunk above `cap` tiles, and the
    reported chunk count is exactly the number of non-empty chunks (= the kernel's grid)"""
    from rfdnet_amd import _lib
    lib = _lib.lib()
    b, e, cnt = C.c_int(), C.c_int(), C.c_int()
    assert lib.rfd_occ_chunk_range_capped(0, n, W, cap, C.byref(b), C.byref(e), C.byref(cnt)) == 0
    total = cnt.value
    pos, sizes = 0, []
    for k in range(total):
        lib.rfd_occ_chunk_range_capped(k, n, W, cap, C.byref(b), C.byref(e), None)
        assert b.value == pos and b.value < e.value <= n, (k, b.value, e.value, pos)
        sizes.append(e.value - b.value)
        pos = e.value
    assert pos == n and max(sizes) <= cap
    lib.rfd_occ_chunk_range_capped(total, n, W, cap, C.byref(b), C.byref(e), None)
    assert b.value == e.value == n
    if n >= 4 * W * cap:
        assert sizes[0] == cap and sizes[-1] == 1
    # cap = 0 is the uncapped schedule
    lib.rfd_occ_chunk_range_capped(3, n, W, 0, C.byref(b), C.byref(e), None)
    b2, e2 = C.c_int(), C.c_int()
    lib.rfd_occ_chunk_range(3, n, W, C.byref(b2), C.byref(e2))
    assert (b.value, e.value) == (b2.value, e2.value)
