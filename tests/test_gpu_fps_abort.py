"""GPU (-m gpu): a multi-workgroup FPS launch whose workgroups cannot be resident together must ABORT -- quickly,
everywhere, with status bit 0 -- and leave the device usable (include/rfd_pointnet2.h
`rfd_fps_set_timeout_ms`).  Reference behaviour matched: a launch that cannot run fails fast (cuda_utils.h:30-39),
it never hangs.

The impossible launch: `rfd_test_hold_cus` parks a workgroup that holds the whole LDS on all but two CUs (it lets go
when told to, or by itself after max_ms); the SA1 shape needs 32 workgroups x 4 waves of ~96 registers = 7 CUs.  (A
CU-masked stream, `hipExtStreamCreateWithCUMask`, was tried first: a mask of 2 CUs was not honoured on this stack --
the launch ran on the whole chip.)"""
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


def _scene():
    pc = synthetic.synthetic_scene(seed=10, n_points=80000)
    return np.ascontiguousarray(pc[None, :, :3])


def _fps(hip, x, m, stream):
    out = torch.zeros(1, m, dtype=torch.int32, device="cuda")
    tmp = torch.empty(1, x.shape[1], device="cuda")
    with torch.cuda.stream(stream):
        rc = hip.lib().furthest_point_sampling_kernel_wrapper(1, x.shape[1], m, x.data_ptr(), tmp.data_ptr(),
                                                              out.data_ptr(), stream.cuda_stream)
    hip.check(rc, "fps")
    return out


class _Hold(object):
    """all but `free` CUs held by rfd_test_hold_cus on a stream of its own, until release() (or max_ms)"""

    def __init__(self, hip, free, max_ms):
        self.hip = hip
        self.flag = torch.zeros(1, dtype=torch.int32, device="cuda")
        self.stream = torch.cuda.Stream()
        self.side = torch.cuda.Stream()
        torch.cuda.synchronize()
        n = hip.lib().rfd_test_hold_cus(free, self.flag.data_ptr(), max_ms, self.stream.cuda_stream)
        assert n > 0, "rfd_test_hold_cus failed: %d" % n
        time.sleep(0.05)                      # every holding workgroup is placed before the launch under test

    def release(self):
        with torch.cuda.stream(self.side):
            self.flag.fill_(1)
        self.stream.synchronize()


@pytest.mark.parametrize("ppt", [10, 5, 40])
def test_fps_that_cannot_be_co_resident_aborts_fast_and_the_device_stays_usable(hip, oracle, short_timeout, ppt):
    p = _scene()
    x = torch.from_numpy(p).cuda()
    ref = oracle.furthest_point_sampling(p, 2048)
    main = torch.cuda.current_stream()
    work = torch.cuda.Stream()
    assert torch.equal(_fps(hip, x, 2048, main).cpu(), torch.from_numpy(ref))       # healthy launch, default geometry
    hip.device_status()
    assert hip.lib().rfd_fps_set_geometry(ppt) >= 0
    # two free CUs hold 10 of the 32 (ppt 10), 14 of the 63 (ppt 5), 2 of the 8 (ppt 40: 350 registers a wave) workgroups
    hold = _Hold(hip, free=2, max_ms=4000)
    try:
        t0 = time.perf_counter()
        out = _fps(hip, x, 2048, work)
        with torch.cuda.stream(work):
            st = hip.stream_status_bits()                   # waits for `work` only: the holders are still there
        dt = time.perf_counter() - t0
        assert st & 1, "a launch that cannot be co-resident must raise status bit 0 (got %d)" % st
        assert dt < 1.0, "abort took %.2f s" % dt
        assert out.cpu().numpy()[0, 0] == 0                 # (whatever follows is garbage or the caller's zero-fill)
        with torch.cuda.stream(work):
            with pytest.raises(hip.RfdHipError, match="furthest point sampling aborted"):
                _fps(hip, x, 2048, work)                    # the same stream, the same exchange region, again
                hip.stream_status()
    finally:
        hold.release()
        hip.lib().rfd_fps_set_geometry(0)
    # the chip is free again: the next launches -- same stream and region, forced and default geometry -- are bit-exact
    hip.lib().rfd_fps_set_geometry(ppt)
    try:
        assert torch.equal(_fps(hip, x, 2048, work).cpu(), torch.from_numpy(ref))
    finally:
        hip.lib().rfd_fps_set_geometry(0)
    assert torch.equal(_fps(hip, x, 2048, work).cpu(), torch.from_numpy(ref))
    assert torch.equal(_fps(hip, x, 2048, main).cpu(), torch.from_numpy(ref))
    with torch.cuda.stream(work):
        assert hip.stream_status_bits() == 0
    hip.device_status()


def test_generate_raises_when_fps_aborts(hip, short_timeout):
    """ISCNet.generate while the chip cannot hold the SA1 launch: RfdHipError, not a hang, not garbage meshes; the same
    call succeeds once the CUs are back."""
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
    hold = _Hold(hip, free=2, max_ms=1500)        # lets go by itself after 1.5 s: the rest of the scene runs at full width
    try:
        t0 = time.perf_counter()
        with torch.no_grad():
            with pytest.raises(hip.RfdHipError, match="furthest point sampling aborted"):
                net.generate({'point_clouds': pc}, selection='all')
        assert time.perf_counter() - t0 < 30.0
    finally:
        hold.release()
    with torch.no_grad():
        _, ids, meshes = net.generate({'point_clouds': pc}, selection='all')
    assert len(meshes) == ids.shape[1] == 256
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
