"""CPU: the N>1 path (scene sharding + the statistics all-gather) with two
processes over gloo on 127.0.0.1 -- the same code bench.py runs over RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from rfdnet_amd import sharding


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_scenes, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ids = sharding.scene_ids_for_rank(n_scenes, rank, world)
    dist.barrier()
    stats = sharding.pack_stats(steps=len(ids), elapsed_s=1.0 + rank, n_meshes=256 * len(ids),
                                n_vertices=100 * rank, n_triangles=7, n_queries=sum(ids),
                                decode_ms=5.0, decode_points=11, decode_launches=3, failed=rank)
    g = sharding.gather_stats(stats, torch.device("cpu"), dist)
    q.put((rank, ids, g))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_scenes", [8, 5])
def test_two_rank_sharding_and_stats_allgather(n_scenes):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_scenes, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    all_ids = sorted(i for _, ids, _ in res for i in ids)
    assert all_ids == list(range(n_scenes))                      # every scene exactly once
    g0, g1 = res[0][2], res[1][2]
    np.testing.assert_array_equal(g0, g1)                        # identical on every rank
    assert g0.shape == (2, len(sharding.STAT_FIELDS))
    np.testing.assert_array_equal(g0[:, 0], [len(res[0][1]), len(res[1][1])])
    value, t_max = sharding.job_throughput(g0)
    assert t_max == 2.0 and value == n_scenes / 2.0              # all scenes over the slowest rank


def test_single_process_needs_no_collective():
    g = sharding.gather_stats(sharding.pack_stats(steps=3, elapsed_s=0.5, n_meshes=1, n_vertices=2,
                                                  n_triangles=3, n_queries=4, decode_ms=5,
                                                  decode_points=6, decode_launches=7, failed=0),
                              torch.device("cpu"), None)
    assert g.shape == (1, 10) and sharding.job_throughput(g) == (6.0, 0.5)
    assert sharding.scene_ids_for_rank(10, 3, 4) == [3, 7]
    with pytest.raises(ValueError):
        sharding.scene_ids_for_rank(4, 4, 4)
    with pytest.raises(KeyError):
        sharding.pack_stats(steps=1)
