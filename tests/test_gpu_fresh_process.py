"""GPU (-m gpu): twenty fresh processes, each running gemm_rows8 / the tile GEMM / both decoders /
the fused SA layer / the fused chains / furthest point sampling ONCE from a cold context against their references (tests/fresh_process_check.py).
Warm loops cannot catch a read of a register whose hand-issued load has not landed yet; the
withdrawn fused ResnetBlockFC kernel of round 1 failed exactly this way in 1 of 14 cold runs."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
N_PROCESSES = 20


def test_hand_scheduled_kernels_from_cold_processes(hip):
    procs, fails = [], []
    # four at a time: concurrent cold contexts also contend for the first page faults
    for base in range(0, N_PROCESSES, 4):
        batch = [subprocess.Popen([sys.executable, os.path.join(HERE, "fresh_process_check.py"), str(s)],
                                  stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
                 for s in range(base, min(base + 4, N_PROCESSES))]
        for s, p in enumerate(batch, start=base):
            out, _ = p.communicate(timeout=600)
            procs.append(s)
            if p.returncode != 0 or ("fresh-process check %d OK" % s) not in out:
                fails.append((s, out[-1500:]))
    assert len(procs) == N_PROCESSES
    assert not fails, "cold-process mismatches in runs %s:\n%s" % ([f[0] for f in fails], fails[0][1])
