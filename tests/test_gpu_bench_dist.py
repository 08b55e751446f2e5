"""GPU (-m gpu): bench.py under torch.distributed.run with the RCCL backend on the one GPU of the box
(world size 1, RFD_BENCH_FORCE_DIST=1): process-group init bound to the device, the barriers bracketing
the timed region and the float64 statistics all-gather run over RCCL exactly as they do for N ranks.
(The N > 1 launcher / sharding logic is covered on CPU with gloo: tests/test_bench_launcher.py.)"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_over_rccl_world_size_one(hip):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, RFD_BENCH_FORCE_DIST="1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                        "--gpus", "1", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-latency"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    # one step = one batch of the scenes in flight per GPU (bench.py's default; 4 since round 4)
    assert out["n_gpus"] == 1 and out["config"]["scenes_failed"] == 0
    assert out["config"]["scenes_done"] == out["config"]["scenes_in_flight_per_gpu"] == 4
    assert out["value"] > 1.0 and out["roofline"]["achieved"] > 100.0
