"""GPU (-m gpu): a multi-workgroup FPS launch that cannot complete its exchange must ABORT -- quickly, everywhere, with
status bit 0 -- and leave the device usable (include/rfd_pointnet2.h `rfd_fps_set_timeout_ms`).
Reference behaviour matched: a launch that cannot run fails fast (cuda_utils.h:30-39), it never hangs.

Two ways in:
  * `rfd_fps_test_phantom_units(1)`: every round also waits for an exchange unit nobody publishes -- exactly what a workgroup
    that is never dispatched looks like to the resident ones.  Deterministic: drives the time-out, the sticky abort word, the
    workgroup-wide exit and the recovery of the same stream / exchange region.
  * `rfd_test_hold_cus`: the real thing -- workgroups holding the whole LDS of all but a few CUs beside the launch.  What the
    hardware then does is its own business (measured in round 5, GPU call 5: with 2 free CUs no workgroup of the grid is
    placed and the launch simply waits for the CUs; with 8 free CUs and the null stream some are placed, time out, and the
    rest leave through the abort word when they arrive), so the property tested is the contract: the launch ends when the
    CUs come back, and its result is either FLAGGED or bit-equal to a healthy launch -- never a hang, never a silent lie.
    (A CU-masked stream, `hipExtStreamCreateWithCUMask`, was tried first: a mask of 2 CUs was not honoured on this stack.)"""
import time

import numpy as np
import pytest
import torch

from rfdnet_amd import synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture()
def short_timeout(hip):
    prev = hip.lib().rfd_fps_set_timeout_ms(40)
    yield 40
    hip.lib().rfd_fps_set_timeout_ms(prev)
    hip.lib().rfd_fps_set_geometry(0)
    hip.lib().rfd_fps_test_phantom_units(0)


def _scene():
    pc = synthetic.synthetic_scene(seed=10, n_points=80000)
    return np.ascontiguousarray(pc[None, :, :3])


def _fps(hip, x, m, stream, wait=True):
    out = torch.zeros(1, m, dtype=torch.int32, device="cuda")
    tmp = torch.empty(1, x.shape[1], device="cuda")
    torch.cuda.current_stream().synchronize()          # the zero-fill above ran on the current stream
    with torch.cuda.stream(stream):
        rc = hip.lib().furthest_point_sampling_kernel_wrapper(1, x.shape[1], m, x.data_ptr(), tmp.data_ptr(),
                                                              out.data_ptr(), stream.cuda_stream)
    hip.check(rc, "fps")
    if wait:
        stream.synchronize()                           # (`.cpu()` copies on the CURRENT stream, not on `stream`)
    return out


@pytest.mark.parametrize("ppt", [10, 5, 40])
def test_fps_exchange_timeout_aborts_the_whole_launch_and_the_stream_recovers(hip, oracle, short_timeout, ppt):
    p = _scene()
    x = torch.from_numpy(p).cuda()
    ref = torch.from_numpy(oracle.furthest_point_sampling(p, 2048))
    main = torch.cuda.current_stream()
    work = torch.cuda.Stream()
    assert torch.equal(_fps(hip, x, 2048, main).cpu(), ref)       # healthy launch, default geometry
    hip.device_status()
    assert hip.lib().rfd_fps_set_geometry(ppt) >= 0
    hip.lib().rfd_fps_test_phantom_units(1)
    try:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = _fps(hip, x, 2048, work, wait=False)
        with torch.cuda.stream(work):
            st = hip.stream_status_bits()                   # waits for `work`
        dt = time.perf_counter() - t0
        assert st & 1, "an exchange that cannot complete must raise status bit 0 (got %d after %.3f s)" % (st, dt)
        assert dt < 1.0, "abort took %.2f s (time-out 40 ms)" % dt
        assert out.cpu().numpy()[0, 0] == 0                 # (whatever follows is garbage or the caller's zero-fill)
        # the hook is one-shot: the call above consumed it, an un-armed launch is healthy again
        assert hip.lib().rfd_fps_test_phantom_units(1) == 0
        with torch.cuda.stream(work):
            with pytest.raises(hip.RfdHipError, match="furthest point sampling aborted"):
                _fps(hip, x, 2048, work, wait=False)        # the same stream, the same exchange region, again
                hip.stream_status()
        assert hip.lib().rfd_fps_test_phantom_units(0) == 0
    finally:
        hip.lib().rfd_fps_test_phantom_units(0)
    # the next launches -- same stream and region, forced and default geometry -- are bit-exact again
    try:
        assert torch.equal(_fps(hip, x, 2048, work).cpu(), ref)
    finally:
        hip.lib().rfd_fps_set_geometry(0)
    assert torch.equal(_fps(hip, x, 2048, work).cpu(), ref)
    assert torch.equal(_fps(hip, x, 2048, main).cpu(), ref)
    with torch.cuda.stream(work):
        assert hip.stream_status_bits() == 0
    hip.device_status()


def test_generate_raises_when_fps_aborts(hip, short_timeout):
    """ISCNet.generate with an SA1 launch that cannot complete: RfdHipError, not a hang, not garbage meshes; the same
    call succeeds afterwards."""
    from rfdnet_amd.iscnet.config import Config
    from rfdnet_amd.iscnet.network import ISCNet
    cfg = Config({'data': {'num_point': 80000}, 'generation': {'resolution_0': 16, 'upsampling_steps': 0}})
    net = ISCNet(cfg)
    synthetic.load_seeded(net, seed=10)
    net = net.cuda().eval()
    pc = torch.from_numpy(synthetic.synthetic_scene(seed=10, n_points=80000)).cuda()[None]
    with torch.no_grad():
        _, ids, meshes = net.generate({'point_clouds': pc}, selection='all')        # builds every lazily packed weight
    assert len(meshes) == ids.shape[1] == 256
    hip.lib().rfd_fps_test_phantom_units(1)
    try:
        t0 = time.perf_counter()
        with torch.no_grad():
            with pytest.raises(hip.RfdHipError, match="furthest point sampling aborted"):
                net.generate({'point_clouds': pc}, selection='all')
        assert time.perf_counter() - t0 < 10.0
    finally:
        hip.lib().rfd_fps_test_phantom_units(0)
    with torch.no_grad():
        _, ids, meshes = net.generate({'point_clouds': pc}, selection='all')
    assert len(meshes) == ids.shape[1] == 256
    hip.device_status()


@pytest.mark.parametrize("free,null_stream", [(2, False), (8, False), (8, True), (40, False)])
def test_fps_beside_cu_holders_never_hangs_and_never_lies(hip, oracle, short_timeout, free, null_stream):
    """the real situation: most CUs are held by somebody else's persistent workgroups (here for at most 1.5 s)"""
    p = _scene()
    x = torch.from_numpy(p).cuda()
    ref = torch.from_numpy(oracle.furthest_point_sampling(p, 2048))
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    hs, side = torch.cuda.Stream(), torch.cuda.Stream()
    work = torch.cuda.default_stream() if null_stream else torch.cuda.Stream()
    out = torch.zeros(1, 2048, dtype=torch.int32, device="cuda")
    tmp = torch.empty(1, 80000, device="cuda")
    torch.cuda.synchronize()
    n = hip.lib().rfd_test_hold_cus(free, flag.data_ptr(), 1500, hs.cuda_stream)
    assert n > 0, n
    time.sleep(0.05)
    t0 = time.perf_counter()
    with torch.cuda.stream(work):
        hip.check(hip.lib().furthest_point_sampling_kernel_wrapper(1, 80000, 2048, x.data_ptr(), tmp.data_ptr(),
                                                                   out.data_ptr(), work.cuda_stream), "fps")
        done = torch.cuda.Event()
        done.record()
    while not done.query() and time.perf_counter() - t0 < 6.0:
        time.sleep(0.002)
    dt = time.perf_counter() - t0
    with torch.cuda.stream(side):
        flag.fill_(1)
    torch.cuda.synchronize()
    assert dt < 3.0, "the launch outlived the CU holders by %.1f s" % (dt - 1.5)
    with torch.cuda.stream(work):
        st = hip.stream_status_bits()
    assert (st & 1) or torch.equal(out.cpu(), ref), "unflagged wrong result beside %d held CUs" % n
    print("free CUs %d, %s stream: done after %.3f s, status %d, %s" % (free, "null" if null_stream else "pool", dt, st,
                                                                        "aborted" if st & 1 else "healthy"))
    # and the device is fine afterwards
    with torch.cuda.stream(work):
        hip.check(hip.lib().furthest_point_sampling_kernel_wrapper(1, 80000, 2048, x.data_ptr(), tmp.data_ptr(),
                                                                   out.data_ptr(), work.cuda_stream), "fps")
        assert hip.stream_status_bits() == 0
    assert torch.equal(out.cpu(), ref)
    hip.device_status()


@pytest.mark.parametrize("ppt", [5, 8, 16, 20, 40])
def test_fps_geometries_are_bit_exact(hip, oracle, ppt):
    """the exchange geometry (rfd_fps_set_geometry: sweeps) never changes a result: SA1 shape incl. the temp scratch"""
    p = _scene()
    ref, rtemp = oracle.furthest_point_sampling(p, 2048, return_temp=True)
    x = torch.from_numpy(p).cuda()
    assert hip.lib().rfd_fps_set_geometry(ppt) >= 0
    try:
        tmp = torch.empty(1, 80000, device="cuda")
        out = torch.zeros(1, 2048, dtype=torch.int32, device="cuda")
        rc = hip.lib().furthest_point_sampling_kernel_wrapper(1, 80000, 2048, x.data_ptr(), tmp.data_ptr(),
                                                              out.data_ptr(), hip.current_stream())
        hip.check(rc, "fps")
        hip.device_status()
    finally:
        hip.lib().rfd_fps_set_geometry(0)
    np.testing.assert_array_equal(out.cpu().numpy(), ref)
    np.testing.assert_array_equal(tmp.cpu().numpy(), rtemp)


@pytest.mark.parametrize("b,n,m,ppt", [(3, 20000, 333, 5), (3, 20000, 333, 16), (2, 5001, 1, 8), (2, 9999, 2500, 10),
                                       (4, 70001, 97, 20), (1, 300000, 64, 32),
                                       # a forced size that leaves ONE workgroup (4096 < n <= 256 ppt) used to
                                       # fail with hipErrorInvalidValue; the automatic geometry now answers
                                       (2, 5000, 77, 32), (1, 9000, 50, 40), (1, 16384, 33, 64)])
def test_forced_geometries_on_odd_shapes_and_batches(hip, oracle, b, n, m, ppt):
    """several scenes per launch (each with its own G exchange units in ONE region), sizes that are no multiple of
    anything, m = 1, duplicates and near-origin points -- against the oracle, geometry forced"""
    rng = np.random.default_rng(n + m + ppt)
    p = rng.uniform(-2, 2, (b, n, 3)).astype(np.float32)
    p[:, rng.integers(0, n, n // 50)] = p[:, rng.integers(0, n, n // 50)]                  # duplicates (exact ties)
    p[:, rng.integers(0, n, 8)] = rng.uniform(-0.01, 0.01, (8, 3)).astype(np.float32)      # skipped points
    ref = oracle.furthest_point_sampling(p, m)
    x = torch.from_numpy(p).cuda()
    out = torch.zeros(b, m, dtype=torch.int32, device="cuda")
    tmp = torch.empty(b, n, device="cuda")
    assert hip.lib().rfd_fps_set_geometry(ppt) >= 0
    try:
        hip.check(hip.lib().furthest_point_sampling_kernel_wrapper(b, n, m, x.data_ptr(), tmp.data_ptr(), out.data_ptr(),
                                                                   hip.current_stream()), "fps")
        hip.device_status()
    finally:
        hip.lib().rfd_fps_set_geometry(0)
    np.testing.assert_array_equal(out.cpu().numpy(), ref)
