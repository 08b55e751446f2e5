"""Multi-GPU plan of the hot path: scenes are independent units (the reference
runs inference with batch 1 and bypasses DataParallel, iscnet/testing.py:61,
demo.py:409), so they shard one-per-GPU with replicated weights and NO
data-path collective.  One process per GPU (`torch.distributed`, backend
"nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).  The only exchange is
a fixed-length float64 statistics vector per rank at the end (latency-bound:
<= 128 B per rank) plus the barrier that brackets the timed region.
"""
import torch

STAT_FIELDS = ("steps", "elapsed_s", "n_meshes", "n_vertices", "n_triangles", "n_queries",
               "decode_ms", "decode_points", "decode_launches")


def scene_ids_for_rank(n_scenes, rank, world_size):
    """Round-robin partition: scene i -> rank i mod world_size."""
    if not (0 <= rank < world_size):
        raise ValueError("rank %d outside world of %d" % (rank, world_size))
    return list(range(rank, n_scenes, world_size))


def pack_stats(**kw):
    missing = [f for f in STAT_FIELDS if f not in kw]
    if missing:
        raise KeyError("missing stats: %s" % missing)
    return [float(kw[f]) for f in STAT_FIELDS]


def gather_stats(stats, device, dist=None):
    """stats: list of len(STAT_FIELDS) floats of THIS rank -> (world, n) float64
    array identical on every rank.  dist=None or world 1 => no communication."""
    if dist is not None and dist.is_initialized() and dist.get_backend() == "gloo":
        device = torch.device("cpu")          # CPU tests / single-GPU dry runs of the N>1 path
    vec = torch.tensor(stats, dtype=torch.float64, device=device)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return vec.view(1, -1).cpu().numpy()
    world = dist.get_world_size()
    out = torch.empty(world * vec.numel(), dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(out, vec)
    return out.view(world, -1).cpu().numpy()


def job_throughput(gathered):
    """Whole-job scenes/s: all ranks' scenes over the SLOWEST rank's time."""
    steps = gathered[:, STAT_FIELDS.index("steps")].sum()
    t_max = gathered[:, STAT_FIELDS.index("elapsed_s")].max()
    return float(steps / t_max), float(t_max)
