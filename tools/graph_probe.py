"""Does a captured HIP graph of the detection stage pay?    One thread, one stream: capture the
stage piece by piece (which piece is not capturable on this stack?), then -- if the whole stage captures -- time eager
against replay on the demo workload's scene.
    python tools/graph_probe.py > profiles/r05_graph_probe.txt"""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rfdnet_amd import _lib, synthetic  # noqa: E402
from rfdnet_amd.iscnet.config import Config  # noqa: E402
from rfdnet_amd.iscnet.network import ISCNet  # noqa: E402
from rfdnet_amd.pointnet2_ops import _ext  # noqa: E402


def try_capture(name, fn, stream):
    """-> (graph or None, outputs)"""
    with torch.cuda.stream(stream):
        for _ in range(2):
            out = fn()                                   # warm: lazily packed weights, library kernel selection
        stream.synchronize()
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g, stream=stream, capture_error_mode="thread_local"):
                out = fn()
            g.replay()
            stream.synchronize()
            print("capture %-28s OK" % name, flush=True)
            return g, out
        except Exception as e:
            print("capture %-28s FAILED: %s: %s" % (name, type(e).__name__, str(e).splitlines()[0][:160]), flush=True)
            try:
                torch.cuda.synchronize()
            except Exception:
                pass
            return None, None


def timed(fn, stream, it=20):
    with torch.cuda.stream(stream):
        for _ in range(3):
            fn()
        stream.synchronize()
        t0 = time.perf_counter()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(it):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / it, 1e3 * (time.perf_counter() - t0) / it


def main():
    cfg = Config({'data': {'num_point': 80000}, 'generation': {'resolution_0': 32, 'upsampling_steps': 0}})
    net = ISCNet(cfg)
    synthetic.load_seeded(net, seed=10)
    net = net.cuda().eval()
    pc = torch.from_numpy(synthetic.synthetic_scene(seed=10, n_points=80000, n_raw=120000)).cuda()[None]
    xyz = pc[..., :3].contiguous()
    s = torch.cuda.Stream()
    print("device %s, torch %s" % (torch.cuda.get_device_name(0), torch.__version__))
    with torch.no_grad():
        try_capture("torch elementwise only", lambda: (pc * 2.0).sum(), s)
        try_capture("FPS SA1 (multi-workgroup)", lambda: _ext.furthest_point_sampling_gather(xyz, 2048), s)
        inds, ctr = _ext.furthest_point_sampling_gather(xyz, 2048)
        try_capture("ball query SA1", lambda: _ext.ball_query(ctr, xyz, 0.2, 64), s)
        try_capture("backbone", lambda: net.backbone(pc, {}), s)
        ep = net.backbone(pc, {})
        try_capture("voting", lambda: net.voting(ep['fp2_xyz'], ep['fp2_features']), s)
        g, out = try_capture("detect (whole stage)", lambda: net.detect(pc), s)
        ms_eager, wall_eager = timed(lambda: net.detect(pc), s)
        print("detect eager : %.3f ms per scene on the stream (HIP events), %.3f ms host wall" % (ms_eager, wall_eager))
        if g is not None:
            ms_g, wall_g = timed(g.replay, s)
            print("detect replay: %.3f ms per scene on the stream (HIP events), %.3f ms host wall" % (ms_g, wall_g))
            ref = net.detect(pc)[0]
            same = all(torch.equal(out[0][k], ref[k]) for k in ("center", "objectness_scores", "sem_cls_scores"))
            print("replayed outputs bit-equal to eager: %s" % same)
    _lib.device_status()


if __name__ == "__main__":
    main()
