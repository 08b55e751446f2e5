"""CPU baseline of the whole hot path (TEST / MEASUREMENT INFRASTRUCTURE, NOT PRODUCT CODE).

Imported only by bench.py's `cpu_baseline` leg.  SURVEY.md §8(d): the reference has no CPU
implementation of its point ops, so the baseline is a PORT --

  * the C oracle (oracle/rfd_oracle.c, OpenMP) for FPS / ball query / grouping / three_nn /
    three_interpolate, MISE and marching cubes,
  * PyTorch-CPU fp32 for every MLP, with the module semantics of the reference restated as
    plain torch layers (random weights: this is a timing, not a parity, leg):
      shared MLPs of the SA / FP layers   pointnet2_modules.py:9-19, 244-255, 395-403
      voting / proposal heads             vote_module.py:34-61, proposal_module.py:85-124
      STN3d / STNkd / PointSeg            pointseg.py:7-166 (un-factored: 1088-channel head)
      ResnetPointnet                      layers.py:340-392 (cat([net, pooled]) materialised)
      DecoderCBatchNorm                   occ_decoder.py:110-123, layers.py:98-107, 226-242
                                          (per proposal, <= 100 000 points per call as
                                          generator.py:123-143 does)
  on the host cores of the box, same scene, bounded samples extrapolated to one scene.
Every stage is reported (`stage_s`), plus 1-core figures for FPS and ball query.

Thread hygiene (round 6).  bench.py pins its GPU-side host threads to the GPU's NUMA node; OpenMP / torch intra-op
workers created under that mask keep it, and 128 workers on one node's cores made the MLP legs 10x slower in rounds
4-5.  The CPU legs therefore run in a FRESH interpreter (`run_isolated`): full affinity mask set before any
library loads, one OpenMP thread per physical core (`OMP_PLACES=cores`, `OMP_PROC_BIND=close`, explicit
`OMP_WAIT_POLICY`), `torch.set_num_threads` = physical cores, one warm-up per leg, and every leg reports its threads,
the size of its affinity mask and its GFLOP/s (`legs`).  torch's bundled libgomp and the oracle's share one SONAME
(`libgomp.so.1`), i.e. ONE runtime and one worker pool in the process.
"""
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np


_AFFINITY_CPUS = None


def physical_cores(cpus=None):
    """one CPU id per physical core among `cpus` (default: this process's affinity mask), from the sibling lists
    the kernel publishes; falls back to the mask itself where sysfs is absent"""
    cpus = sorted(os.sched_getaffinity(0) if cpus is None else cpus)
    seen, firsts = set(), []
    for c in cpus:
        try:
            with open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c) as f:
                txt = f.read().strip()
            sib = []
            for part in txt.split(","):
                lo, _, hi = part.partition("-")
                sib += list(range(int(lo), int(hi or lo) + 1))
            key = tuple(sorted(sib))
        except (OSError, ValueError):
            key = (c,)
        if key not in seen:
            seen.add(key)
            firsts.append(c)
    return firsts


def widest_affinity():
    """every CPU this process may be given: the online set if the cgroup allows it, else the current mask"""
    cur = os.sched_getaffinity(0)
    try:
        os.sched_setaffinity(0, range(os.cpu_count() or 1))
        full = os.sched_getaffinity(0)
        os.sched_setaffinity(0, cur)
        return sorted(full)
    except OSError:
        return sorted(cur)


def child_env(n_threads, wait_policy=None):
    """environment of the isolated CPU-baseline process: everything that sizes or places a thread pool is explicit"""
    env = {k: v for k, v in os.environ.items()
           if not k.startswith(("OMP_", "GOMP_", "KMP_", "MKL_")) and k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update({"OMP_NUM_THREADS": str(n_threads), "MKL_NUM_THREADS": str(n_threads),
                "OMP_PLACES": "cores", "OMP_PROC_BIND": "close", "OMP_DYNAMIC": "false", "MKL_DYNAMIC": "false",
                "OMP_WAIT_POLICY": wait_policy or os.environ.get("RFD_CPU_BASELINE_WAIT", "passive"),
                "HIP_VISIBLE_DEVICES": "", "ROCR_VISIBLE_DEVICES": ""})          # a CPU process: no GPU context
    return env


def run_isolated(kind="scene", cpus=None, timeout_s=1800, threads=None, wait_policy=None, **kwargs):
    """bench.py's entry: run `run(**kwargs)` (kind "scene") or `run_decoder_only(**kwargs)` (kind "decoder") in a fresh
    interpreter with the full affinity mask `cpus` and explicit OpenMP settings; -> the `cpu_baseline` object."""
    cpus = sorted(cpus) if cpus else widest_affinity()
    n_threads = int(threads) if threads else len(physical_cores(cpus))
    grids = kwargs.pop("scene_grids", None)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory(prefix="rfd_cpu_baseline_") as tmp:
        if grids:
            kwargs["scene_grids_npz"] = os.path.join(tmp, "grids.npz")
            np.savez(kwargs["scene_grids_npz"], *[np.asarray(g) for g in grids])
        job = os.path.join(tmp, "job.json")
        with open(job, "w") as f:
            json.dump({"kind": kind, "cpus": cpus, "threads": n_threads, "kwargs": kwargs}, f)
        r = subprocess.run([sys.executable, "-m", "oracle.cpu_baseline", job], cwd=root, env=child_env(n_threads, wait_policy),
                           capture_output=True, text=True, timeout=timeout_s)
    if r.returncode != 0:
        raise RuntimeError("cpu_baseline child failed (rc %d):\n%s" % (r.returncode, (r.stdout + r.stderr)[-4000:]))
    return json.loads(r.stdout.strip().splitlines()[-1])


def sockets(cpus):
    """number of CPU packages among `cpus` (1 where sysfs does not say)"""
    ids = set()
    for c in cpus:
        try:
            with open("/sys/devices/system/cpu/cpu%d/topology/physical_package_id" % c) as f:
                ids.add(int(f.read()))
        except (OSError, ValueError):
            pass
    return max(1, len(ids))


def run_best(kind="scene", cpus=None, **kwargs):
    """bench.py's entry.  The isolated run on every physical core and -- on a multi-socket host -- on ONE socket's
    cores (`OMP_PROC_BIND=close` fills socket 0 first; measured on the 2 x 64-core MI355X host: the one-socket run
    is 10-20 % FASTER, the legs' buffers being first-touched by one thread, profiles/r06_cpu_baseline_repeat.txt).
    The faster one is the baseline (`cores` = the threads it used); the other is kept under `alternatives`."""
    cpus = sorted(cpus) if cpus else widest_affinity()
    phys = len(physical_cores(cpus))
    plans = [phys]
    n_sock = sockets(cpus)
    if n_sock > 1 and phys // n_sock >= 1:
        plans.append(phys // n_sock)
    outs = [run_isolated(kind, cpus=cpus, threads=t, **dict(kwargs)) for t in plans]
    best = max(outs, key=lambda o: o["value"])
    best["alternatives"] = [{"threads": o["cores"], "value": o["value"], "unit": o["unit"],
                             "stage_s": o.get("stage_s")} for o in outs if o is not best]
    return best


def _probe():
    """where the worker threads of this process may run, after one parallel torch op and one OpenMP oracle call:
    the union of every thread's allowed CPUs (the defect of rounds 4-5 was workers confined to one NUMA node)"""
    import torch
    from oracle import oracle
    torch.mm(torch.randn(512, 512), torch.randn(512, 512))
    oracle.furthest_point_sampling(np.random.default_rng(0).random((1, 4096, 3)).astype(np.float32), 8)
    union, n = set(), 0
    for tid in os.listdir("/proc/self/task"):
        try:
            union |= os.sched_getaffinity(int(tid))
            n += 1
        except OSError:
            pass
    return {"threads_alive": n, "worker_cpus_union": len(union), "omp_threads": oracle.num_threads()}


def _child_main(job_path):
    with open(job_path) as f:
        job = json.load(f)
    try:                                                    # before numpy / torch / libgomp load: workers inherit it
        os.sched_setaffinity(0, job["cpus"])
    except OSError:
        pass
    global _AFFINITY_CPUS
    _AFFINITY_CPUS = len(os.sched_getaffinity(0))           # the process's mask; OMP_PROC_BIND then narrows each thread to its core
    import torch
    torch.set_num_threads(job["threads"])
    kw = dict(job["kwargs"])
    path = kw.pop("scene_grids_npz", None)
    if path:
        with np.load(path) as z:
            kw["scene_grids"] = [z[k] for k in z.files]
    out = {"scene": run, "decoder": run_decoder_only, "probe": _probe}[job["kind"]](**kw)
    out["isolation"] = {"fresh_process": True, "affinity_cpus": _AFFINITY_CPUS,
                        "physical_cores": job["threads"], "torch_threads": torch.get_num_threads(),
                        "env": {k: os.environ.get(k) for k in ("OMP_NUM_THREADS", "OMP_PLACES", "OMP_PROC_BIND",
                                                               "OMP_WAIT_POLICY", "MKL_NUM_THREADS")}}
    print(json.dumps(out))


def _mlp2d(torch, dims):
    nn = torch.nn
    layers = []
    for a, b in zip(dims[:-1], dims[1:]):
        layers += [nn.Conv2d(a, b, 1, bias=False), nn.BatchNorm2d(b), nn.ReLU(inplace=True)]
    return nn.Sequential(*layers).eval()


def _mlp1d(torch, dims, last_plain=False):
    nn = torch.nn
    layers = []
    n = len(dims) - 1
    for i, (a, b) in enumerate(zip(dims[:-1], dims[1:])):
        layers.append(nn.Conv1d(a, b, 1))
        if not (last_plain and i == n - 1):
            layers += [nn.BatchNorm1d(b), nn.ReLU(inplace=True)]
    return nn.Sequential(*layers).eval()


class _TNet(object):
    def __init__(self, torch, k_in, k_out):
        nn = torch.nn
        self.torch = torch
        self.conv = _mlp1d(torch, [k_in, 64, 128, 1024])
        self.fc = nn.Sequential(nn.Linear(1024, 512), nn.BatchNorm1d(512), nn.ReLU(),
                                nn.Linear(512, 256), nn.BatchNorm1d(256), nn.ReLU(),
                                nn.Linear(256, k_out * k_out)).eval()
        self.k = k_out

    def __call__(self, x):
        g = self.conv(x).max(2)[0]
        return (self.fc(g) + self.torch.eye(self.k).view(1, -1)).view(-1, self.k, self.k)


class _PointSeg(object):
    """pointseg.py:85-166 with feature_transform=True, global_feat=False"""

    def __init__(self, torch, channel):
        nn = torch.nn
        self.torch = torch
        self.stn = _TNet(torch, channel, 3)
        self.conv1 = _mlp1d(torch, [channel, 64])
        self.fstn = _TNet(torch, 64, 64)
        self.conv2 = _mlp1d(torch, [64, 128])
        self.conv3 = nn.Sequential(nn.Conv1d(128, 1024, 1), nn.BatchNorm1d(1024)).eval()
        self.head = _mlp1d(torch, [1088, 512, 256, 128, 2], last_plain=True)

    def __call__(self, x):                               # (B, D, N)
        torch = self.torch
        B, D, N = x.shape
        trans = self.stn(x)
        xt = x.transpose(2, 1)
        xt = torch.cat([torch.bmm(xt[..., :3], trans), xt[..., 3:]], dim=2)
        h = self.conv1(xt.transpose(2, 1))
        tf = self.fstn(h)
        h = torch.bmm(h.transpose(2, 1), tf).transpose(2, 1)
        g = self.conv3(self.conv2(h)).max(2, keepdim=True)[0]
        y = self.head(torch.cat([g.repeat(1, 1, N), h], 1))
        return torch.log_softmax(y.transpose(2, 1).reshape(-1, 2), -1).view(B, N, 2)


class _ResnetPointnet(object):
    def __init__(self, torch, dim, hidden, c_dim):
        nn = torch.nn
        self.torch = torch
        self.fc_pos = nn.Linear(dim, 2 * hidden)
        self.blocks = [(nn.Linear(2 * hidden, hidden), nn.Linear(hidden, hidden),
                        nn.Linear(2 * hidden, hidden, bias=False)) for _ in range(5)]
        self.fc_c = nn.Linear(hidden, c_dim)

    def _block(self, i, x):
        torch = self.torch
        fc0, fc1, sc = self.blocks[i]
        ax = torch.relu(x)
        return sc(ax) + fc1(torch.relu(fc0(ax)))

    def __call__(self, p):                               # (B, T, dim)
        torch = self.torch
        net = self._block(0, self.fc_pos(p))
        for i in range(1, 5):
            pooled = net.max(1, keepdim=True)[0].expand(net.size())
            net = self._block(i, torch.cat([net, pooled], 2))
        return self.fc_c(torch.relu(net.max(1)[0]))


class _Decoder(object):
    """DecoderCBatchNorm as the module computes it (nothing folded)."""

    def __init__(self, torch, c_dim=512, hidden=256, z_dim=32):
        nn = torch.nn
        self.torch = torch
        self.fc_p = nn.Conv1d(3, hidden, 1)
        self.fc_z = nn.Linear(z_dim, hidden)

        def cbn():
            return (nn.Conv1d(c_dim, hidden, 1), nn.Conv1d(c_dim, hidden, 1),
                    nn.BatchNorm1d(hidden, affine=False).eval())
        self.blocks = [(cbn(), nn.Conv1d(hidden, hidden, 1), cbn(), nn.Conv1d(hidden, hidden, 1))
                       for _ in range(5)]
        self.bn = cbn()
        self.fc_out = nn.Conv1d(hidden, 1, 1)

    @staticmethod
    def _cbn(m, x, c):
        g, b, bn = m
        return g(c) * bn(x) + b(c)

    def __call__(self, p, z, c):                          # p (1,T,3), z (1,32), c (1,512)
        torch = self.torch
        c = c.unsqueeze(2)
        net = self.fc_p(p.transpose(1, 2)) + self.fc_z(z).unsqueeze(2)
        for b0, fc0, b1, fc1 in self.blocks:
            h = fc0(torch.relu(self._cbn(b0, net, c)))
            net = net + fc1(torch.relu(self._cbn(b1, h, c)))
        return self.fc_out(torch.relu(self._cbn(self.bn, net, c))).squeeze(1)


def _timed(fn, reps=1):
    fn()                                                   # one warm-up per leg (thread pool, allocator, oneDNN primitives)
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps


def _chain_flop(n, dims):
    """2 * n * sum(a * b) over consecutive widths: a pointwise MLP over n positions"""
    return 2.0 * n * sum(a * b for a, b in zip(dims[:-1], dims[1:]))


def pointseg_flop(n, channel=4):
    """_PointSeg on one proposal of n points (pointseg.py:85-166)"""
    tnet = lambda k_in, k_out: _chain_flop(n, [k_in, 64, 128, 1024]) + _chain_flop(1, [1024, 512, 256, k_out * k_out])
    return (tnet(channel, 3) + _chain_flop(n, [channel, 64]) + tnet(64, 64) + 2.0 * n * 64 * 64
            + _chain_flop(n, [64, 128, 1024]) + _chain_flop(n, [1088, 512, 256, 128, 2]))


def resnet_pointnet_flop(n, dim=132, hidden=512, c_dim=512):
    """_ResnetPointnet on one proposal of n points (layers.py:340-392)"""
    return (_chain_flop(n, [dim, 2 * hidden]) + 5 * 2.0 * n * (2 * hidden * hidden + hidden * hidden + 2 * hidden * hidden)
            + _chain_flop(1, [hidden, c_dim]))


DECODER_FLOP_PER_POINT = 1312768.0


def run(points=80000, resolution0=32, upsampling_steps=1, n_queries_per_scene=None, n_prop=256,
        budget_s=20.0, scene_grids=None, threshold_logit=0.0, min_skip_sample=32):
    """-> the `cpu_baseline` object of bench.py's JSON line.
    scene_grids: value grids (R+1)^3 of a few proposals of THE SCENE (the CPU path's own, computed by bench.py's parity
    leg with the oracle decoder): the octree and marching-cubes legs then run on the scene's fields instead of an
    analytic sphere -- the octree is replayed with the grid as its field (same queries, same subdivisions)."""
    import torch
    from oracle import oracle
    from rfdnet_amd import synthetic
    oracle.build()
    cores = torch.get_num_threads()                        # set by the caller (_child_main: one per physical core)
    oracle.set_num_threads(cores)
    t, sample, gflop = {}, {}, {}
    pc = synthetic.synthetic_scene(seed=10, n_points=points, n_raw=120000 if points > 60000 else 30000)
    xyz = np.ascontiguousarray(pc[None, :, :3])

    # ---- point ops: C oracle, all cores (full scene) -------------------------------------
    t0 = time.perf_counter()
    i1 = oracle.furthest_point_sampling(xyz, 2048)
    x1 = xyz[:, i1[0]]
    i2 = oracle.furthest_point_sampling(x1, 1024); x2 = x1[:, i2[0]]
    i3 = oracle.furthest_point_sampling(x2, 512); x3 = x2[:, i3[0]]
    i4 = oracle.furthest_point_sampling(x3, 256); x4 = x3[:, i4[0]]
    oracle.furthest_point_sampling(x2, 256)
    t['fps'] = time.perf_counter() - t0
    t0 = time.perf_counter()
    idx1 = oracle.ball_query(x1, xyz, 0.2, 64)
    idx2 = oracle.ball_query(x2, x1, 0.4, 32)
    oracle.ball_query(x3, x2, 0.8, 16)
    oracle.ball_query(x4, x3, 1.2, 16)
    oracle.ball_query(x2[:, :256], x2, 0.3, 16)
    idx_sp = oracle.ball_query(x4 + np.float32(0.05), xyz, 1.0, 1024)          # skip propagation, K=256
    t['ball_query'] = time.perf_counter() - t0
    t0 = time.perf_counter()
    oracle.group_points(np.ascontiguousarray(xyz.transpose(0, 2, 1)), idx1)
    oracle.group_points(np.random.default_rng(0).normal(size=(1, 128, 2048)).astype(np.float32), idx2)
    oracle.group_points(np.ascontiguousarray(pc[None].transpose(0, 2, 1)), idx_sp)
    d2, i3nn = oracle.three_nn(x2, x3)
    oracle.three_interpolate(np.zeros((1, 256, 512), np.float32), i3nn, d2)
    t['group_interp'] = time.perf_counter() - t0
    # 1-core figures (reported, not part of the total): FPS rounds and ball-query centres scale linearly
    one = {}
    oracle.set_num_threads(1)
    try:
        t0 = time.perf_counter()
        oracle.furthest_point_sampling(xyz, 129)
        one['fps_sa1_s'] = (time.perf_counter() - t0) * 2047 / 128
        t0 = time.perf_counter()
        oracle.ball_query(x1[:, :128], xyz, 0.2, 64)
        one['ball_query_sa1_s'] = (time.perf_counter() - t0) * 2048 / 128
    finally:
        oracle.set_num_threads(cores)

    # ---- MLPs: PyTorch-CPU fp32, module semantics, all cores ------------------------------
    with torch.no_grad():
        sa = [(_mlp2d(torch, [4, 64, 64, 128]), torch.randn(1, 4, 2048, 64)),
              (_mlp2d(torch, [131, 128, 128, 256]), torch.randn(1, 131, 1024, 32)),
              (_mlp2d(torch, [259, 128, 128, 256]), torch.randn(1, 259, 512, 16)),
              (_mlp2d(torch, [259, 128, 128, 256]), torch.randn(1, 259, 256, 16))]
        fp = [(_mlp2d(torch, [512, 256, 256]), torch.randn(1, 512, 512, 1)),
              (_mlp2d(torch, [512, 256, 256]), torch.randn(1, 512, 1024, 1))]

        def backbone():
            for m, x in sa:
                torch.nn.functional.max_pool2d(m(x.clone()), kernel_size=[1, x.size(3)])
            for m, x in fp:
                m(x.clone())
        t['mlp_backbone'] = _timed(backbone, reps=3)
        gflop['mlp_backbone'] = 1e-9 * (
            _chain_flop(2048 * 64, [4, 64, 64, 128]) + _chain_flop(1024 * 32, [131, 128, 128, 256])
            + _chain_flop(512 * 16, [259, 128, 128, 256]) + _chain_flop(256 * 16, [259, 128, 128, 256])
            + _chain_flop(512, [512, 256, 256]) + _chain_flop(1024, [512, 256, 256]))
        vote = _mlp1d(torch, [256, 256, 256, 259], last_plain=True)
        agg = _mlp2d(torch, [259, 128, 128, 128])
        head = _mlp1d(torch, [128, 128, 128, 69], last_plain=True)
        xv, xa, xh = torch.randn(1, 256, 1024), torch.randn(1, 259, 256, 16), torch.randn(1, 128, 256)

        def vote_prop():
            vote(xv)
            torch.nn.functional.max_pool2d(agg(xa.clone()), kernel_size=[1, 16])
            head(xh)
        t['mlp_vote_proposal'] = _timed(vote_prop, reps=3)
        gflop['mlp_vote_proposal'] = 1e-9 * (_chain_flop(1024, [256, 256, 256, 259]) + _chain_flop(256 * 16, [259, 128, 128, 128])
                                             + _chain_flop(256, [128, 128, 128, 69]))

        # skip propagation nets on a sample of proposals
        seg, enc = _PointSeg(torch, 4), _ResnetPointnet(torch, 4 + 128, 512, 512)
        ks = min(n_prop, min_skip_sample)                    # >= 32 proposals: one batch, as skip_propagation.py does
        xs, xe = torch.randn(ks, 4, 1024), torch.randn(ks, 1024, 132)
        ts = _timed(lambda: (seg(xs), enc(xe)))
        if ts < 0.15 * budget_s / 4 and ks < n_prop:         # fast box: widen the sample
            ks = int(min(n_prop, max(ks, ks * (0.15 * budget_s / 2) / ts)))
            xs, xe = torch.randn(ks, 4, 1024), torch.randn(ks, 1024, 132)
            ts = _timed(lambda: (seg(xs), enc(xe)))
        t['skip_propagation_nets'] = ts * n_prop / ks
        gflop['skip_propagation_nets'] = 1e-9 * n_prop * (pointseg_flop(1024) + resnet_pointnet_flop(1024))
        sample['skip_propagation_nets'] = "%d of %d proposals" % (ks, n_prop)

        # decoder: the calls one proposal makes (generator.py:99-143) -- round 0 = the (res0+1)^3 lattice in one call,
        # then one call per upsampling round with that proposal's share of the scene's remaining queries (<= 100 000
        # points per call, generator.py:129-141) -- on a sample of proposals
        if n_queries_per_scene is None:
            n_queries_per_scene = n_prop * ((resolution0 + 1) ** 3 if upsampling_steps else resolution0 ** 3)
        per_prop = max(1, int(round(n_queries_per_scene / n_prop)))
        if upsampling_steps:
            n0 = min(per_prop, (resolution0 + 1) ** 3)
            rest = per_prop - n0
            calls = [n0] + [rest // upsampling_steps] * upsampling_steps if rest >= upsampling_steps else [n0]
        else:
            calls = [per_prop]
        calls = [m for c_ in calls for m in [100000] * (c_ // 100000) + ([c_ % 100000] if c_ % 100000 else [])]
        n_call = sum(calls)
        dec = _Decoder(torch)
        z, c = torch.zeros(1, 32), torch.randn(1, 512)
        pts = (torch.rand(1, max(calls), 3) - 0.5) * 1.1

        def dec_props(k):
            t0 = time.perf_counter()
            for _ in range(k):
                for m in calls:
                    dec(pts[:, :m], z, c)
            return time.perf_counter() - t0
        dec_props(1)                                        # warm-up: every call shape once
        t_cal = max(dec_props(1), 1e-4)
        k_s = int(min(n_prop, max(2, (0.2 * budget_s) / t_cal)))
        t_dec = dec_props(k_s)
        n_s = k_s * n_call
    # the same decoder as the OpenMP C restatement (oracle_decoder_cbn: fp32, all cores, the same calls)
    # -- BOTH legs are reported, the faster one counts
    from rfdnet_amd.iscnet.occ_decoder import DecoderCBatchNorm
    dmod = DecoderCBatchNorm(dim=3, z_dim=32, c_dim=512, hidden_size=256)
    synthetic.load_seeded(dmod, 3)
    blob = oracle.decoder_param_blob({k: v.numpy() for k, v in dmod.state_dict().items()})
    zc, cc = np.zeros((1, 32), np.float32), np.random.default_rng(1).normal(0, 1, (1, 512)).astype(np.float32)
    pc_ = ((np.random.default_rng(2).random((1, max(calls), 3)) - 0.5) * 1.1).astype(np.float32)

    def cdec_props(k):
        t0 = time.perf_counter()
        for _ in range(k):
            for m in calls:
                oracle.decoder_cbn(blob, np.ascontiguousarray(pc_[:, :m]), zc, cc)
        return time.perf_counter() - t0
    cdec_props(1)
    tc_cal = max(cdec_props(1), 1e-4)
    k_c = int(min(n_prop, max(2, (0.2 * budget_s) / tc_cal)))
    t_cdec = cdec_props(k_c)
    n_c = k_c * n_call
    legs = {'decoder_torch_module': t_dec * n_queries_per_scene / n_s,
            'decoder_c_oracle_openmp': t_cdec * n_queries_per_scene / n_c}
    best = min(legs, key=legs.get)
    t['decoder'] = legs[best]
    sample['decoder'] = ("%s leg; calls per proposal %s (torch module: %d proposals = %d points in %.1f s = %.0f points/s; "
                         "OpenMP C restatement: %d proposals = %d points in %.1f s = %.0f points/s), of %d query points"
                         % (best, calls, k_s, n_s, t_dec, n_s / t_dec, k_c, n_c, t_cdec, n_c / t_cdec,
                            n_queries_per_scene))

    # ---- MISE octree + marching cubes: C oracle on a sample of proposals --------------------
    grids = []
    if scene_grids and upsampling_steps > 0:
        n_m = len(scene_grids)
        t0 = time.perf_counter()
        for g in scene_grids:
            g = np.asarray(g, np.float64)
            m = oracle.MISE(resolution0, upsampling_steps, threshold_logit)
            q = m.query()
            while q.shape[0]:
                m.update(q, g[q[:, 0], q[:, 1], q[:, 2]])
                q = m.query()
            grids.append(m.to_dense())
        t['mise_octree'] = (time.perf_counter() - t0) / n_m * n_prop
        sample['mise_octree'] = "%d of %d proposals of the scene (octree replayed on the CPU path's value grids)" % (n_m, n_prop)
        thr_mc, what = threshold_logit, "the scene's value grids"
    elif upsampling_steps > 0:
        n_m = 4
        t0 = time.perf_counter()
        for _ in range(n_m):
            m = oracle.MISE(resolution0, upsampling_steps, 0.0)
            q = m.query()
            while q.shape[0]:
                cc = q.astype(np.float64) / m.resolution - 0.5
                m.update(q, 0.35 - np.sqrt((cc ** 2).sum(-1)))
                q = m.query()
            grids.append(m.to_dense())
        t['mise_octree'] = (time.perf_counter() - t0) / n_m * n_prop
        sample['mise_octree'] = "%d of %d proposals (analytic sphere field)" % (n_m, n_prop)
        thr_mc, what = 0.0, "sphere grids"
    else:
        n_m = 4
        n = resolution0
        idx = np.stack(np.meshgrid(*[np.linspace(-0.5, 0.5, n)] * 3, indexing="ij"), -1)
        grids = [0.35 - np.sqrt((idx ** 2).sum(-1))] * n_m
        t['mise_octree'] = 0.0
        thr_mc, what = 0.0, "sphere grids"
    t0 = time.perf_counter()
    faces = 0
    for g in grids:
        faces += len(oracle.extract_mesh(g, thr_mc)[1])
    t['marching_cubes'] = (time.perf_counter() - t0) / n_m * n_prop
    sample['marching_cubes'] = "%d of %d proposals (%d^3, %s, %d faces per proposal)" % (
        n_m, n_prop, grids[0].shape[0], what, faces // max(n_m, 1))

    total = sum(t.values())
    gflop['decoder'] = 1e-9 * DECODER_FLOP_PER_POINT * n_queries_per_scene
    legs_out = {k: {"s": round(v, 4), "threads": cores, "affinity_cpus": _AFFINITY_CPUS or len(os.sched_getaffinity(0)),
                    "gflop": round(gflop[k], 2) if k in gflop else None,
                    "gflops": round(gflop[k] / v, 1) if k in gflop and v > 0 else None} for k, v in t.items()}
    legs_out["decoder"].update({"torch_points_per_s": round(n_s / t_dec), "c_points_per_s": round(n_c / t_cdec),
                                "torch_gflops": round(1e-9 * DECODER_FLOP_PER_POINT * n_s / t_dec, 1),
                                "c_gflops": round(1e-9 * DECODER_FLOP_PER_POINT * n_c / t_cdec, 1)})
    return {"value": 1.0 / total, "unit": "scenes/s", "cores": cores, "kind": "port", "legs": legs_out,
            "sample": ("one scene of the same workload on %d host cores: C oracle point ops on the full scene; "
                       "PyTorch-CPU fp32 MLPs of the backbone / voting / proposal stages in full; "
                       "skip-propagation nets on %s; decoder on %s; MISE octree on %s; marching cubes on %s "
                       "-- each extrapolated linearly to the scene"
                       % (cores, sample['skip_propagation_nets'], sample['decoder'],
                          sample.get('mise_octree', 'n/a (dense grid)'), sample['marching_cubes'])),
            "stage_s": {k: round(v, 4) for k, v in t.items()},
            "decoder_legs_s": {k: round(v, 3) for k, v in legs.items()},
            "one_core_s": {k: round(v, 3) for k, v in one.items()},
            "scene_s": round(total, 3)}


def run_decoder_only(n_total=256 * 262144, budget_s=15.0):
    """configs[2] on the host: the restated DecoderCBatchNorm on a bounded sample of the query
    points, points/s."""
    import torch
    cores = torch.get_num_threads()                        # set by the caller (_child_main: one per physical core)
    dec = _Decoder(torch)
    z, c = torch.zeros(1, 32), torch.randn(1, 512)
    with torch.no_grad():
        def dec_time(n):
            p = (torch.rand(1, n, 3) - 0.5) * 1.1
            t0 = time.perf_counter()
            dec(p, z, c)
            return time.perf_counter() - t0
        dec_time(4096)
        t_cal = max(dec_time(16384), 1e-4)
        n_s = int(min(100000, max(16384, 16384 * budget_s / t_cal)))
        t = dec_time(n_s)
    return {"value": n_s / t, "unit": "points/s", "cores": cores, "kind": "port",
            "sample": "PyTorch-CPU fp32 DecoderCBatchNorm (module semantics) on %d of %d query points of one "
                      "proposal (%.1f s measured)" % (n_s, n_total, t)}


if __name__ == "__main__":
    _child_main(sys.argv[1])
