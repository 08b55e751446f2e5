"""The host-CPU baseline (oracle/cpu_baseline.py, measurement infrastructure) is reproducible and not confined.

Rounds 4-5 printed a baseline ~4x too slow: bench.py pins itself to the GPU's NUMA node, the OpenMP / torch workers
created under that mask kept it, and 128 workers shared one node's cores.  The legs now run in a fresh interpreter
with the full mask; these tests pin the two properties that matter: a pinned PARENT does not confine the child's
workers, and two back-to-back runs of a leg agree."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_physical_cores_counts_siblings_once():
    from oracle import cpu_baseline
    cores = cpu_baseline.physical_cores()
    assert 1 <= len(cores) <= len(os.sched_getaffinity(0))
    assert len(set(cores)) == len(cores) and set(cores) <= os.sched_getaffinity(0)
    assert cpu_baseline.physical_cores([min(os.sched_getaffinity(0))]) == [min(os.sched_getaffinity(0))]


def test_child_env_is_explicit():
    from oracle import cpu_baseline
    os.environ["OMP_NUM_THREADS"] = "3"
    try:
        env = cpu_baseline.child_env(5)
    finally:
        del os.environ["OMP_NUM_THREADS"]
    assert env["OMP_NUM_THREADS"] == "5" and env["MKL_NUM_THREADS"] == "5"
    assert env["OMP_PLACES"] == "cores" and env["OMP_PROC_BIND"] == "close" and env["OMP_WAIT_POLICY"] in ("passive", "active")
    assert "RANK" not in env and env["HIP_VISIBLE_DEVICES"] == ""


@pytest.mark.skipif(len(os.sched_getaffinity(0)) < 2, reason="needs two CPUs")
def test_pinned_parent_does_not_confine_the_workers():
    """the regression itself: the parent sits on ONE cpu (as bench.py sits on one NUMA node) with its own OpenMP pool
    already created there; the isolated child still spreads its workers over the whole mask"""
    full = sorted(os.sched_getaffinity(0))
    code = ("import os, json\n"
            "full = %r\n"
            "os.sched_setaffinity(0, {full[0]})\n"
            "import torch; torch.mm(torch.randn(256, 256), torch.randn(256, 256))     # parent pool, born pinned\n"
            "from oracle import cpu_baseline\n"
            "print(json.dumps(cpu_baseline.run_isolated('probe', cpus=full)))\n" % (full,))
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    from oracle import cpu_baseline
    n_phys = len(cpu_baseline.physical_cores(full))
    assert out["isolation"]["affinity_cpus"] == len(full)
    assert out["isolation"]["torch_threads"] == n_phys and out["omp_threads"] == n_phys
    assert out["worker_cpus_union"] >= n_phys, out
    assert out["isolation"]["env"]["OMP_PROC_BIND"] == "close"


def test_two_back_to_back_runs_agree():
    """the same leg run again agrees within 1.5x (the round-5 lines swung 12x across runs).  Three
    runs, and the two CLOSEST of each leg must agree: the build container's 8 cores are shared, one run may catch a
    noisy neighbour -- a sick thread pool is slow every time."""
    from oracle import cpu_baseline
    runs = [cpu_baseline.run_isolated("scene", budget_s=4.0, n_prop=64, min_skip_sample=8) for _ in range(3)]
    for leg in ("skip_propagation_nets", "decoder"):
        v = sorted(r["stage_s"][leg] for r in runs)
        ratio = min(v[1] / v[0], v[2] / v[1])
        assert ratio < 1.5, (leg, v)
    for out in runs:
        assert out["isolation"]["fresh_process"] and out["legs"]["decoder"]["gflops"] > 0
        assert set(out["legs"]) == set(out["stage_s"])
        assert all(v["threads"] == out["cores"] for v in out["legs"].values())
