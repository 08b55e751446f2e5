"""Multi-GPU plan of the hot path: scenes are independent units (the reference
runs inference with batch 1 and bypasses DataParallel, iscnet/testing.py:61,
demo.py:409), so they shard one-per-GPU with replicated weights and NO
data-path collective.  One process per GPU (`torch.distributed`, backend
"nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).  The only exchange is
a fixed-length float64 statistics vector per rank at the end (latency-bound:
<= 128 B per rank) plus the barrier that brackets the timed region.
"""
import os
import socket
import subprocess
import sys

import torch

STAT_FIELDS = ("steps", "elapsed_s", "n_meshes", "n_vertices", "n_triangles", "n_queries",
               "decode_ms", "decode_points", "decode_launches", "failed")


def rank_env():
    """(rank, local_rank, world) of this process as torch.distributed.run exports them."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def launched():
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _cpulist(text):
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus += list(range(int(lo), int(hi or lo) + 1))
    return cpus


def numa_cpus_for_gpu(index, sys_root="/sys"):
    """Host CPUs of the NUMA node GPU `index` hangs off (KFD topology: the index-th node with SIMDs -> PCI
    domain / location id -> /sys/bus/pci/devices/<bdf>/numa_node -> node<N>/cpulist).  None when the topology cannot
    be read or the GPU reports no node (-1): the caller then leaves the affinity alone."""
    try:
        base = os.path.join(sys_root, "class", "kfd", "kfd", "topology", "nodes")
        gpus = []
        for n in sorted(os.listdir(base), key=int):
            props = {}
            try:
                with open(os.path.join(base, n, "properties")) as fh:
                    for line in fh:
                        k, _, v = line.strip().partition(" ")
                        props[k] = v
            except OSError:
                continue        # a container sees only its own GPUs' nodes: the others are unreadable (EPERM) or empty
            if int(props.get("simd_count", "0")) > 0:
                gpus.append(props)
        p = gpus[index]
        loc, dom = int(p["location_id"]), int(p.get("domain", "0"))
        bdf = "%04x:%02x:%02x.%d" % (dom, (loc >> 8) & 0xff, (loc >> 3) & 0x1f, loc & 7)
        with open(os.path.join(sys_root, "bus", "pci", "devices", bdf, "numa_node")) as fh:
            node = int(fh.read().strip())
        if node < 0:
            return None
        with open(os.path.join(sys_root, "devices", "system", "node", "node%d" % node, "cpulist")) as fh:
            cpus = _cpulist(fh.read())
        return cpus or None
    except (OSError, ValueError, KeyError, IndexError):
        return None


def numa_cpus_for_bdf(bdf, sys_root="/sys"):
    """Host CPUs of the NUMA node of the PCI device `bdf` ("0000:f1:00.0"), None when unknown.  The answer HIP itself
    gives for a device (torch.cuda.get_device_properties(i).pci_*) -- it does not depend on how KFD nodes, readable or
    not, line up with HIP's device numbering, so a rank checks its pre-initialisation pinning
    against it once the runtime is up (bench.py)."""
    try:
        with open(os.path.join(sys_root, "bus", "pci", "devices", bdf, "numa_node")) as fh:
            node = int(fh.read().strip())
        if node < 0:
            return None
        with open(os.path.join(sys_root, "devices", "system", "node", "node%d" % node, "cpulist")) as fh:
            return _cpulist(fh.read()) or None
    except (OSError, ValueError):
        return None


def device_bdf(props):
    """PCI address of a torch device-properties object, None when this torch build does not expose it"""
    try:
        return "%04x:%02x:%02x.0" % (int(props.pci_domain_id), int(props.pci_bus_id), int(props.pci_device_id))
    except (AttributeError, TypeError, ValueError):
        return None


def visible_device_index(local_rank, env=None):
    """Physical (KFD order) index of the GPU that HIP calls device `local_rank`, honouring HIP_VISIBLE_DEVICES /
    ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES remapping; None when a list is set that cannot be translated (UUIDs,
    out of range): the caller then skips pinning rather than pin to another GPU's NUMA node."""
    env = os.environ if env is None else env
    idx = local_rank
    # ROCR_VISIBLE_DEVICES filters what the runtime enumerates; HIP_VISIBLE_DEVICES (CUDA_VISIBLE_DEVICES is its alias,
    # used only when the HIP variable is absent) indexes into that list: translate from the inside out
    hip_var = "HIP_VISIBLE_DEVICES" if (env.get("HIP_VISIBLE_DEVICES") or "").strip() else "CUDA_VISIBLE_DEVICES"
    for var in (hip_var, "ROCR_VISIBLE_DEVICES"):
        v = env.get(var)
        if v is None or v.strip() == "":
            continue
        try:
            ids = [int(x) for x in v.split(",") if x.strip() != ""]
            idx = ids[idx]
        except (ValueError, IndexError):
            return None
    return idx


def pin_cpus_for_rank(local_rank, env=None, sys_root="/sys"):
    """The CPUs rank `local_rank` may be pinned to: its GPU's NUMA node INTERSECTED with this process's allowed set
    (docker --cpuset-cpus, the k8s static CPU manager: a cpulist outside it makes sched_setaffinity fail with EINVAL).
    None = leave the affinity alone (no topology, remapping that cannot be translated, empty intersection)."""
    phys = visible_device_index(local_rank, env)
    if phys is None:
        return None
    cpus = numa_cpus_for_gpu(phys, sys_root)
    if not cpus:
        return None
    try:
        allowed = os.sched_getaffinity(0)
    except (AttributeError, OSError):
        return None
    cpus = sorted(set(cpus) & set(allowed))
    return cpus or None


def _pin(cpus):
    """preexec_fn of a rank: never raise (an exception here becomes SubprocessError in the parent and aborts the launch)"""
    try:
        os.sched_setaffinity(0, cpus)
    except (OSError, ValueError):
        pass


def _stop(procs):
    for p in procs:
        if p.poll() is None:
            p.terminate()
    for p in procs:
        try:
            p.wait(10)
        except subprocess.TimeoutExpired:
            p.kill()


def launch_local_ranks(script, argv, nproc, env=None, timeout=None):
    """One process per GPU of THIS node: re-executes `script argv` nproc times with the
    environment torch.distributed.run would set (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR=127.0.0.1 / MASTER_PORT), rank 0 inheriting stdout, each rank's host threads
    pinned to the CPUs of its GPU's NUMA node when the topology says which AND this process may
    use them (pin_cpus_for_rank; RFD_PIN_NUMA=0 disables it).  Returns the worst exit code; if a
    rank fails -- or cannot even be spawned -- the others are terminated (no orphan holding a GPU)."""
    base = dict(os.environ)
    base.update(env or {})
    base.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), WORLD_SIZE=str(nproc),
                LOCAL_WORLD_SIZE=str(nproc))
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC (RCCL across processes)
    procs = []
    pin = base.get("RFD_PIN_NUMA", "1") != "0" and hasattr(os, "sched_setaffinity")
    for r in range(nproc):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        # host threads of rank r (its three scene workers, the pinned-memory copies) stay on the NUMA node of GPU r:
        # the mesh D2H copies and kernel launches do not cross the socket interconnect
        cpus = pin_cpus_for_rank(r, base) if pin else None
        pre = (lambda c=cpus: _pin(c)) if cpus else None
        try:
            procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=e,
                                          stdout=None if r == 0 else subprocess.DEVNULL, preexec_fn=pre))
        except (OSError, subprocess.SubprocessError):
            _stop(procs)               # a rank that cannot be spawned takes the ranks already started with it
            raise
    import time
    t_end = None if timeout is None else time.time() + timeout
    rc = 0
    alive = list(procs)
    while alive:
        for p in list(alive):
            c = p.poll()
            if c is not None:
                alive.remove(p)
                if c != 0:
                    rc = rc or c
        if rc or (t_end is not None and time.time() > t_end):
            _stop(alive)
            return rc or 124
        time.sleep(0.05)
    return rc


PREFLIGHT_FIELDS = ("rank", "device_index", "visible_devices", "pinned_cpus", "allowed_cpus", "numa_node_matches",
                    "hw_queues", "free_gib", "library_ok")


def preflight(rank, local_rank, world, stub=False, one_device=False, deadline_s=55.0, env=None, allowed_cpus=None,
              need_gib=8.0):
    """`bench.py --preflight`: can this job start?  Run by EVERY rank (the driver's first `--gpus 8` run is unattended):
      1. local checks any rank can evaluate by itself, BEFORE the rendezvous, so that a mis-sized job fails on every rank
         at once instead of hanging in it: ranks per node vs visible devices, the HIP library loads and is a gfx950 build,
         the device is a gfx950 part;
      2. the rendezvous + ONE all-gather of a small per-rank vector over the job's backend (RCCL over xGMI on GPUs, gloo
         in the CPU tests) -- the collective the benchmark itself ends with -- with the process group's own time-out
         inside the deadline;
      3. a per-rank report: device index, CPUs pinned (now) / allowed (`allowed_cpus`: the mask BEFORE the caller pinned
         itself, so the report shows whether pinning happened), whether the NUMA node of the pinning is the one HIP's
         PCI address says, GPU_MAX_HW_QUEUES, free HBM.
    Any failure prints ONE line `preflight FAILED [rank r]: <what> -- <what to do>` to stderr and exits non-zero; a
    watchdog ends the process with the same kind of line if anything blocks past `deadline_s`.
    Returns the gathered (world, len(PREFLIGHT_FIELDS)) array (rank 0 prints it as JSON)."""
    import datetime
    import threading
    env = os.environ if env is None else env

    def fail(what, fix, code=2):
        sys.stderr.write("preflight FAILED [rank %d]: %s -- %s\n" % (rank, what, fix))
        sys.stderr.flush()
        os._exit(code)

    dog = threading.Timer(deadline_s, lambda: fail(
        "no answer within %.0f s (rendezvous or collective blocked)" % deadline_s,
        "check that all %d ranks started, MASTER_ADDR=127.0.0.1 / MASTER_PORT agree, and HSA_ENABLE_IPC_MODE_LEGACY=0 "
        "is exported" % world, 3))
    dog.daemon = True
    dog.start()
    local_world = int(env.get("LOCAL_WORLD_SIZE", world))
    fake = env.get("RFD_PREFLIGHT_FAKE_DEVICES")
    if stub:
        n_dev = int(fake) if fake else local_world
    else:
        n_dev = torch.cuda.device_count()
    need = 1 if one_device else local_world
    if n_dev < need:
        fail("%d ranks on this node but %d visible GPU(s)" % (local_world, n_dev),
             "launch with --nproc-per-node %d / --gpus %d, or widen HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES"
             % (max(n_dev, 1), max(n_dev, 1)))
    dev_index = 0 if one_device else local_rank
    lib_ok, free_gib, numa_ok = 1.0, -1.0, -1.0
    pinned = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else -1
    allowed = pinned if allowed_cpus is None else int(allowed_cpus)
    device = torch.device("cpu")
    if not stub:
        try:
            from . import _lib
            arch = _lib.lib().rfd_build_arch().decode()
        except Exception as e:
            fail("librfd_hip.so does not load (%s)" % e, "run `python -c 'import __graft_entry__ as g; g.build()'`")
        torch.cuda.set_device(dev_index)
        device = torch.device("cuda", dev_index)
        props = torch.cuda.get_device_properties(dev_index)
        gcn = getattr(props, "gcnArchName", "")
        if arch not in gcn:
            fail("device %d is %s (%s), the library is built for %s" % (dev_index, props.name, gcn, arch),
                 "run on MI355X (gfx950) -- the kernels are written for that part only")
        free, _ = torch.cuda.mem_get_info(dev_index)
        free_gib = free / 2.0 ** 30
        if free_gib < need_gib:              # per config (bench.py HBM_NEED_GIB): a shared GPU is fine for the small ones
            fail("only %.1f GiB of HBM free on device %d, this configuration needs %.0f" % (free_gib, dev_index, need_gib),
                 "another process holds the GPU (rocm-smi --showpids), or pick a smaller --config / --in-flight")
        want = numa_cpus_for_bdf(device_bdf(props) or "")
        if want and hasattr(os, "sched_getaffinity"):
            mine = os.sched_getaffinity(0)
            numa_ok = 1.0 if set(mine) <= set(want) else 0.0
    vec = [float(rank), float(dev_index), float(n_dev), float(pinned), float(allowed), numa_ok,
           float(env.get("GPU_MAX_HW_QUEUES", "-1") or -1), free_gib, lib_ok]
    gathered = None
    if world > 1 or env.get("RFD_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        backend = "gloo" if (stub or one_device) else "nccl"
        kw = {"device_id": device} if backend == "nccl" else {}
        try:
            dist.init_process_group(backend=backend, timeout=datetime.timedelta(seconds=max(10.0, deadline_s - 10.0)), **kw)
            t = torch.tensor(vec, dtype=torch.float64, device=device if backend == "nccl" else "cpu")
            out = torch.empty(world * t.numel(), dtype=torch.float64, device=t.device)
            dist.all_gather_into_tensor(out, t)
            gathered = out.view(world, -1).cpu().numpy()
            dist.barrier()
            dist.destroy_process_group()
        except Exception as e:
            fail("the %s rendezvous / all-gather failed (%s: %s)" % (backend, type(e).__name__, str(e)[:160]),
                 "check MASTER_ADDR=127.0.0.1, a free MASTER_PORT, HSA_ENABLE_IPC_MODE_LEGACY=0, and that no rank died above")
        ranks = sorted(int(r) for r in gathered[:, 0])
        if ranks != list(range(world)):
            fail("the all-gather returned ranks %s" % ranks, "every rank must run the same command line (RANK 0..%d)" % (world - 1))
    else:
        import numpy as np
        gathered = np.asarray([vec])
    dog.cancel()
    return gathered


def scene_ids_for_rank(n_scenes, rank, world_size):
    """Round-robin partition: scene i -> rank i mod world_size."""
    if not (0 <= rank < world_size):
        raise ValueError("rank %d outside world of %d" % (rank, world_size))
    return list(range(rank, n_scenes, world_size))


def scene_ids_for_worker(rank_ids, worker, n_workers, per_pass=1):
    """Split a rank's scenes over its in-flight workers in passes of `per_pass` scenes:
    pass j (scenes [j*per_pass, (j+1)*per_pass)) -> worker j mod n_workers."""
    passes = [rank_ids[i:i + per_pass] for i in range(0, len(rank_ids), per_pass)]
    return passes[worker::n_workers]


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
    if dist is None or not dist.is_initialized():
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
