"""GPU (-m gpu): a multi-workgroup FPS launch whose workgroups cannot be resident together must ABORT -- quickly,
everywhere, with status bit 0 -- and leave the device usable (include/rfd_pointnet2.h
`rfd_fps_set_timeout_ms`).  Reference behaviour matched: a launch that cannot run fails fast (cuda_utils.h:30-39),
it never hangs.

The impossible launch: a stream confined to 2 CUs (hipExtStreamCreateWithCUMask) -- the SA1 shape needs 32
workgroups x 4 waves of ~107 registers = 8 CUs -- also with more (63) and fewer, fatter (8) exchange units."""
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


@pytest.mark.parametrize("ppt", [10, 5, 40])
def test_fps_that_cannot_be_co_resident_aborts_fast_and_the_device_stays_usable(hip, oracle, short_timeout, ppt):
    p = _scene()
    x = torch.from_numpy(p).cuda()
    ref = oracle.furthest_point_sampling(p, 2048)
    main = torch.cuda.current_stream()
    assert torch.equal(_fps(hip, x, 2048, main).cpu(), torch.from_numpy(ref))       # healthy launch, default geometry
    hip.device_status()
    assert hip.lib().rfd_fps_set_geometry(ppt) >= 0
    # ppt 40 -> 8 workgroups: one CU holds one (350 registers a wave), so confine THAT launch to 4 CUs
    n_cus = 4 if ppt == 40 else 2
    masked = hip.cu_masked_stream(0, n_cus)
    try:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = _fps(hip, x, 2048, masked)
        with torch.cuda.stream(masked):
            st = hip.stream_status_bits()                   # waits for the masked stream only
        dt = time.perf_counter() - t0
        assert st & 1, "a launch that cannot be co-resident must raise status bit 0 (got %d)" % st
        assert dt < 1.0, "abort took %.2f s" % dt
        assert out.cpu().numpy()[0, 0] == 0                 # (whatever follows is garbage or the caller's zero-fill)
        with torch.cuda.stream(masked):
            with pytest.raises(hip.RfdHipError, match="furthest point sampling aborted"):
                _fps(hip, x, 2048, masked)
                hip.stream_status()
        # the next launches on the same device -- other stream, same exchange region mechanism -- are bit-exact again,
        # with the forced geometry and with the default one
        assert torch.equal(_fps(hip, x, 2048, main).cpu(), torch.from_numpy(ref))
        hip.lib().rfd_fps_set_geometry(0)
        assert torch.equal(_fps(hip, x, 2048, main).cpu(), torch.from_numpy(ref))
        hip.device_status()
    finally:
        hip.lib().rfd_fps_set_geometry(0)
        hip.check(hip.lib().rfd_stream_destroy(masked.cuda_stream), "rfd_stream_destroy")


def test_generate_raises_when_fps_aborts(hip, short_timeout):
    """ISCNet.generate on a stream that cannot hold the SA1 launch: RfdHipError, not a hang, not garbage meshes."""
    from rfdnet_amd.iscnet.config import Config
    from rfdnet_amd.iscnet.network import ISCNet
    cfg = Config({'data': {'num_point': 80000}, 'generation': {'resolution_0': 16, 'upsampling_steps': 0}})
    net = ISCNet(cfg)
    synthetic.load_seeded(net, seed=10)
    net = net.cuda().eval()
    pc = torch.from_numpy(synthetic.synthetic_scene(seed=10, n_points=80000)).cuda()[None]
    masked = hip.cu_masked_stream(0, 2)
    try:
        t0 = time.perf_counter()
        with torch.cuda.stream(masked), torch.no_grad():
            with pytest.raises(hip.RfdHipError, match="furthest point sampling aborted"):
                net.generate({'point_clouds': pc}, selection='all')
        assert time.perf_counter() - t0 < 60.0
        with torch.no_grad():
            _, ids, meshes = net.generate({'point_clouds': pc}, selection='all')    # default stream: fine
        assert len(meshes) == ids.shape[1] == 256
        hip.device_status()
    finally:
        hip.check(hip.lib().rfd_stream_destroy(masked.cuda_stream), "rfd_stream_destroy")


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
