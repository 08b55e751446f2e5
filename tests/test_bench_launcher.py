"""CPU: `python bench.py --gpus 2` spawns its own two ranks (rfdnet_amd/sharding.launch_local_ranks),
shards scenes with sharding.scene_ids_for_rank, counts a failing scene instead of dying, gathers the
statistics over gloo and prints ONE JSON line with n_gpus = 2.  RFD_BENCH_STUB=1 swaps the HIP scene for
a CPU stand-in -- this exercises the launcher and the accounting, not the kernels."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(argv, **env):
    e = dict(os.environ, RFD_BENCH_STUB="1", **env)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        if k not in env:
            e.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=e, capture_output=True,
                       text=True, timeout=300)
    return p


def test_self_launch_two_ranks_and_failure_accounting():
    p = run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--in-flight", "2"],
                  RFD_BENCH_STUB_FAIL="9")
    assert p.returncode == 0, p.stderr
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout                     # rank 0 only, one line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1
    c = out["config"]
    assert c["scenes_per_step"] == 4                     # 2 ranks x 2 in flight
    assert c["scenes_failed"] == 1 and c["scenes_done"] == 3 * 4 - 1      # scene 9 is in the timed region
    assert "scene(s) [9] failed" in p.stderr
    assert out["scaling"] == "weak" and out["unit"] == "scenes/s" and out["value"] > 0
    assert "cpu_baseline" not in out                     # rank-0-at-N=1 only


def test_single_process_default():
    p = run_bench(["--steps", "2", "--warmup", "0", "--no-cpu-baseline"])
    assert p.returncode == 0, p.stderr
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert out["n_gpus"] == 1 and out["config"]["scenes_failed"] == 0 and out["config"]["scenes_done"] == 6


def test_world_size_mismatch_fails_loudly():
    p = run_bench(["--gpus", "4"], WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                  MASTER_PORT="29999")
    assert p.returncode != 0 and "WORLD_SIZE=2" in p.stderr


def test_every_scene_is_run_exactly_once_across_ranks_and_workers():
    from rfdnet_amd import sharding
    world, S, NB, steps, warm = 4, 3, 2, 5, 2
    per_step = S * NB * world
    seen = []
    for rank in range(world):
        lo, hi = warm * per_step, (warm + steps) * per_step
        mine = [i for i in sharding.scene_ids_for_rank(hi, rank, world) if i >= lo]
        assert all(i % world == rank for i in mine)
        for w in range(S):
            for ids in sharding.scene_ids_for_worker(mine, w, S, NB):
                assert len(ids) == NB
                seen += ids
    assert sorted(seen) == list(range(warm * per_step, (warm + steps) * per_step))


def test_failing_rank_takes_the_job_down(tmp_path):
    from rfdnet_amd import sharding
    script = tmp_path / "r.py"
    script.write_text("import os, sys, time\n"
                      "if os.environ['RANK'] == '1': sys.exit(3)\n"
                      "time.sleep(60)\n")
    rc = sharding.launch_local_ranks(str(script), [], 2)
    assert rc == 3
