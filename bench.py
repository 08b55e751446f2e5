#!/usr/bin/env python
"""Contract benchmark: scenes/sec end-to-end reconstruction (ScanNet-like scene,
80 000 points, 256 proposals, 64^3 MISE) on N MI355X.

  python bench.py --gpus N --steps K --warmup W          (N > 1: spawns its own N ranks)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N \\
         --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one pass of the hot path over one batch of scenes per GPU:
  backbone (FPS / ball query / grouping / three_nn HIP kernels + 1x1-conv MLPs)
  -> voting -> vote aggregation + proposal head -> all 256 proposals
  -> skip propagation (K x 80 000 ball query, PointSeg, ResnetPointnet)
  -> batched 64^3 MISE with the fused MFMA occupancy decoder
  -> batched marching cubes -> meshes copied to host memory.
Scenes shard across GPUs (scene i -> rank i mod N, rfdnet_amd/sharding.py; weak scaling); the
only collective is the all-gather of a small per-rank statistics vector.  Inputs (scenes,
weights) are synthetic and seeded, resident in HBM before the timed region.  A scene that
raises is counted as failed, the run goes on.  Prints ONE JSON line on rank 0.

--config selects the workload (BASELINE.json `configs`):
  headline (default)  configs[1]: 80 000 points, 256 proposals, 64^3 MISE
  mise128             configs[4] per GPU: 128^3 MISE (resolution_0 32, upsampling_steps 2)
  dense32             configs[0]: 40 000 points (sampled WITH replacement), dense 32^3 grid
  stress              configs[2]: decoder only, 256 proposals x 262 144 query points (f16x3 timed; the single-pass f16
                      mode and the matrix-pipe-busy counter ride in `roofline`)
  demo                NOT a BASELINE config: the reference's `main.py --mode demo` workload (demo.py:200-276, ISCNet_test.yaml):
                      80 000 points, proposals selected by objectness + 3-D NMS (about a dozen survive), dense 32^3 grids

--scenes N   sweep N scenes in the timed region (scene i -> rank i mod world), e.g. `--config mise128 --scenes 311` =
             BASELINE configs[4] (the 311 scans of datasets/splits/fullscan/scannetv2_test.json) in one command
--preflight  check the job before anything is built: rank count vs visible devices, a collective all-gather, the NUMA /
             affinity / hardware-queue settings of every rank; one actionable line and a non-zero exit within 60 s
"""
import argparse
import json
import os
import sys
import threading
import time

# Every scene in flight owns a compute stream and a copy stream.  ROCm maps HIP streams onto GPU_MAX_HW_QUEUES hardware
# queues (default 4) in the order of their first use, and two streams sharing one queue serialise -- a 30-ms mesh blit then
# holds up another scene's kernels.  Measured (profiles/r04_hw_queues.txt): 2 queues -8 %, 8-16 queues +1-2 % at three scenes
# in flight and +3-5 % at four (which does not pay with 4 queues).  Read by the HIP runtime when it initialises, so it is set
# before anything touches the GPU; an explicit setting of the caller wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from rfdnet_amd import sharding  # noqa: E402  (no torch.cuda use at import)

FLOP_PER_QUERY = 2 * 3 * 256 + 10 * 2 * 256 * 256 + 2 * 256        # 1 312 768
MFMA_PEAK_TFLOPS = 2500.0      # dense bf16/f16 MFMA, MI355X_MICROARCH.md
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "decoder_traffic.json")

CONFIGS = {
    "headline": dict(points=80000, resolution0=32, upsampling_steps=1, raw=120000,
                     name="configs[1]: one ScanNet-like scene per forward pass, %(points)d points, 256 proposals, "
                          "%(res)d^3 MISE (res0=%(resolution0)d, steps=%(upsampling_steps)d), meshes to host"),
    "mise128": dict(points=80000, resolution0=32, upsampling_steps=2, raw=120000,
                    name="configs[4] on one GPU: %(points)d points, 256 proposals, %(res)d^3 MISE "
                         "(res0=%(resolution0)d, steps=%(upsampling_steps)d) + marching cubes, meshes to host"),
    "dense32": dict(points=40000, resolution0=32, upsampling_steps=0, raw=30000,
                    name="configs[0] on the GPU: %(points)d points sampled with replacement from a %(raw)d-vertex "
                         "scan, 256 proposals, dense %(resolution0)d^3 occupancy grid, meshes to host"),
    "stress": dict(points=0, resolution0=0, upsampling_steps=0, raw=0,
                   name="configs[2]: occupancy decoder stress, 256 proposals x 262 144 query points"),
    "demo": dict(points=80000, resolution0=32, upsampling_steps=0, raw=120000, selection="nms",
                 name="the reference's demo workload (main.py --mode demo; NOT a BASELINE config): %(points)d points, "
                      "proposals kept by objectness > 0.5 + class-aware 3-D NMS (demo.py:223-234), dense "
                      "%(resolution0)d^3 occupancy grids (ISCNet_test.yaml:61-63), meshes to host"),
}
MFMA_BUSY_FILE = os.path.join(ROOT, "profiles", "decoder_mfma_busy.json")


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)       # 32 scenes at four in flight: ~1.6 s timed
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="headline")
    ap.add_argument("--points", type=int, default=None)
    ap.add_argument("--resolution0", type=int, default=None)
    ap.add_argument("--upsampling-steps", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--stats-out", type=str, default=None,
                    help="write per-scene statistics (stage ms from HIP events on the scene's stream, query points, "
                         "vertices, triangles, failed) of the timed region as JSON to this path (rank r > 0: "
                         "<path>.rank<r>); the reference prints per-scene time only (demo.py:408-411)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the per-op roofline measurements after the timed region (grouping, furthest-point sampling, "
                         "ball query): profiling runs that want the scene's own kernels only")
    ap.add_argument("--no-latency", action="store_true",
                    help="skip the one-scene-at-a-time pass after the timed region")
    ap.add_argument("--in-flight", type=int, default=4,
                    help="scenes reconstructed concurrently per GPU (one host thread + HIP stream + model "
                         "replica each); a step = one such batch")
    ap.add_argument("--batch", type=int, default=1,
                    help="scenes per forward pass (batched through every kernel: the latency-bound FPS rounds "
                         "and the ~300 small launches are shared by the batch)")
    ap.add_argument("--blit-round", type=int, default=0,
                    help="MISE round whose decode launch the previous scene's device-to-host mesh copy is released "
                         "behind (0 = the first, longest launch; -1 = at once, when the meshes are complete)")
    ap.add_argument("--mode", choices=["f16x3", "f16x1"], default="f16x3",
                    help="decoder arithmetic; f16x3 is the parity mode (1e-4 on logits)")
    ap.add_argument("--scenes", type=int, default=None,
                    help="sweep: this many scenes in the timed region over the whole job (scene i -> rank i mod world); "
                         "--steps is then derived (ceil(scenes / scenes per step)) and only reported")
    ap.add_argument("--preflight", action="store_true",
                    help="check ranks / devices / collective / affinity and exit (0 = the job can start)")
    ap.add_argument("--demo-keep", type=int, default=16,
                    help="--config demo: the objectness bias of the seeded head is shifted so that this many of scene "
                         "0's 256 proposals pass the 0.5 threshold (NMS and empty-box removal then thin them out)")
    args = ap.parse_args(argv)
    c = CONFIGS[args.config]
    for k in ("points", "resolution0", "upsampling_steps"):
        if getattr(args, k) is None:
            setattr(args, k, c[k])
    args.raw = c["raw"]
    args.selection = c.get("selection", "all")
    if args.config == "stress":
        args.in_flight = 1
    if args.scenes is not None:
        if args.scenes <= 0 or args.config == "stress":
            ap.error("--scenes needs a positive count and a scene configuration")
        per_step = max(1, args.in_flight) * max(1, args.batch) * max(1, args.gpus)
        args.steps = -(-args.scenes // per_step)
    return args


# ------------------------------------------------------------------ HIP backend ---
DECODE_LOCK = threading.Lock()


class DecodeTimer(object):
    """HIP-event timing of every decoder launch on the stream it runs on.  The decoder owns the whole chip (one
    persistent workgroup per CU), so with several scenes in flight its launches execute one after the other anyway;
    to make each event-bracketed duration the kernel's own, a launch's stream first WAITS (on the device) for the end
    event of the previous decoder launch of any stream, then records its start event.  No host thread blocks for it
    (round 2 held a host lock across `e1.synchronize()` inside the timed region); the lock below only covers the
    enqueue.  Durations are read after the timed region's final synchronisation."""

    last_end = None            # end event of the most recent decoder launch on this device (any stream)

    def __init__(self, dec):
        import torch
        self.dec = dec
        self.records = []
        self.enabled = False
        self.local = threading.local()   # .round = MISE round of the calling worker's next launch (the round hook sets it)
        self._orig = dec.decode_tiles

        def wrapped(pts, tile_prop, *a, **k):
            with DECODE_LOCK:
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                if DecodeTimer.last_end is not None:
                    torch.cuda.current_stream().wait_event(DecodeTimer.last_end)
                e0.record()
                out = self._orig(pts, tile_prop, *a, **k)
                e1.record()
                DecodeTimer.last_end = e1
            if self.enabled:
                self.records.append((int(tile_prop.shape[0]) * 128, e0, e1, getattr(self.local, "round", 0)))
            return out
        dec.decode_tiles = wrapped

    def totals(self):
        return (sum(e0.elapsed_time(e1) for _, e0, e1, _ in self.records), sum(n for n, _, _, _ in self.records),
                len(self.records))

    def by_round(self):
        """{round: [ms, padded points, launches]} of the recorded launches"""
        out = {}
        for n, e0, e1, r in self.records:
            a = out.setdefault(r, [0.0, 0, 0])
            a[0] += e0.elapsed_time(e1)
            a[1] += n
            a[2] += 1
        return out


class MeshSink(object):
    """Meshes end up in (pinned) host memory like the reference's trimesh objects
    (generator.py:181-183).  The D2H copy of scene i runs on its own stream and
    overlaps the reconstruction of scene i+1 (two host buffer sets, alternating);
    everything is drained before the clock stops."""

    def __init__(self, device):
        import torch
        self.stream = torch.cuda.Stream(device)
        self.bufs = [[None, None], [None, None]]
        self.turn = 0
        self.pending = None

    def push(self, v, f):
        """Queue the meshes of the scene just finished; the copy is STARTED by
        start_pending() when the next scene launches its first (longest) decode:
        small kernels run 3-10x slower with a PCIe copy in flight (the
        multi-workgroup FPS, which exchanges 8-byte granules through memory every
        round, 70 % slower); the MFMA-bound decoder does not care."""
        self.pending = (v, f)

    def start_pending(self):
        import torch
        if self.pending is None:
            return
        v, f = self.pending
        self.pending = None
        nv, nt = int(v.shape[0]), int(f.shape[0])
        if not nv:
            return
        for b in self.bufs:          # pinned allocation is slow: size BOTH sets at first use (warm-up)
            if b[0] is None or b[0].shape[0] < nv:
                b[0] = torch.empty(int(nv * 1.5) + 1, 3, dtype=torch.float64).pin_memory()
            if b[1] is None or b[1].shape[0] < nt:
                b[1] = torch.empty(int(nt * 1.5) + 1, 3, dtype=torch.int32).pin_memory()
        hb = self.bufs[self.turn]
        self.turn ^= 1
        done = torch.cuda.Event()
        done.record()                                   # meshes complete on the compute stream
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(done)
            hb[0][:nv].copy_(v, non_blocking=True)
            hb[1][:nt].copy_(f, non_blocking=True)
        v.record_stream(self.stream)
        f.record_stream(self.stream)

    def drain(self):
        self.start_pending()
        self.stream.synchronize()


class HipBackend(object):
    """The product path on one GPU: `in_flight` workers (HIP stream + mesh sink + a view of the ONE
    read-only model with its own MISE / mesh state each) over a pool of HBM-resident synthetic scenes."""

    name = "hip"

    def __init__(self, args, rank, local_rank, world):
        import numpy as np
        import torch
        from rfdnet_amd import _lib, synthetic
        assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
        # RFD_BENCH_ONE_DEVICE=1: dry run of the N>1 code path on a 1-GPU box (all ranks
        # on cuda:0, gloo for the statistics exchange) -- never used for reported numbers
        self.one_dev = os.environ.get("RFD_BENCH_ONE_DEVICE") == "1"
        dev_index = 0 if self.one_dev else local_rank
        torch.cuda.set_device(dev_index)
        self.device = torch.device("cuda", dev_index)
        self.dist_backend = "gloo" if self.one_dev else "nccl"          # "nccl" = RCCL over xGMI
        self.args, self.torch, self._lib = args, torch, _lib
        self.S = max(1, args.in_flight)
        self.NB = max(1, args.batch)
        self.record_scenes, self.scene_records = False, []
        self.round_points = {}
        # ONE set of weights / packed weight streams per GPU; every in-flight worker gets a view with its own
        # generator state (round 3 built a full replica per worker)
        self.net = self._build_net()
        self.nets = [self.net] + [self.net.worker_view() for _ in range(self.S - 1)]
        self.timers = [DecodeTimer(self.net.completion.decoder)]
        self.streams = [torch.cuda.Stream(self.device) for _ in range(self.S)]
        self.sinks = [MeshSink(self.device) for _ in range(self.S)]
        # scene pool: global scene id i -> seed 10 + (i mod pool), pool a multiple of the world size so
        # that a rank always meets the same resident scenes (its residue class)
        self.pool = world * 2 * self.S * self.NB
        if getattr(args, "scenes", None):
            # a sweep meets (up to 320) DISTINCT scenes, like the split it stands in for, not eight of them over and over
            total = args.warmup * self.S * self.NB * world + args.scenes
            self.pool = world * (-(-min(total, 320) // world))
        self.world = world
        self.scenes = {}
        for i in sharding.scene_ids_for_rank(self.pool, rank, world):
            pc = synthetic.synthetic_scene(seed=10 + i, n_points=args.points, n_raw=args.raw)
            self.scenes[i] = torch.from_numpy(pc).to(self.device)
        torch.cuda.synchronize()
        self.demo_calibration = None
        if args.selection == "nms":
            self.demo_calibration = self.calibrate_objectness(args.demo_keep)
        # (the lazily built caches of the SHARED model -- packed weight streams, folded BatchNorms -- are built under
        # _lib.BUILD_LOCK and published before they are stored, so the workers may start together.  A priming scene run
        # from THIS thread was tried instead and cost 8 % scenes/s for the rest of the process: profiles/r04_numa_prime.txt)

    def _build_net(self):
        from rfdnet_amd import synthetic
        from rfdnet_amd.iscnet import occ_decoder
        from rfdnet_amd.iscnet.config import Config
        from rfdnet_amd.iscnet.network import ISCNet
        a = self.args
        cfg = Config({'data': {'num_point': a.points},
                      'generation': {'resolution_0': a.resolution0, 'upsampling_steps': a.upsampling_steps}})
        net = ISCNet(cfg)
        synthetic.load_seeded(net, seed=10)           # the reference's seed (ISCNet_test.yaml:5)
        net = net.to(self.device).eval()
        net.completion.decoder.mode = occ_decoder.MODE_F16X3 if a.mode == "f16x3" else occ_decoder.MODE_F16X1
        self.placeholder_sizes = bool(getattr(cfg.dataset_config, "placeholder_sizes", False))
        if a.selection == "nms" and self.placeholder_sizes:
            # no datasets/scannet/scannet_means.npz on the box: the synthetic run opts in (flagged in `config`)
            cfg.eval_overrides = dict(cfg.eval_overrides or {}, allow_placeholder_sizes=True)
        return net

    def calibrate_objectness(self, keep):
        """--config demo: with seeded random weights the objectness head is noise around zero, so the reference's
        selection (probability > dump_threshold, then NMS) would keep ~half of the 256 proposals or none.  Shift the
        bias of the `object` logit so that exactly `keep` proposals of scene 0 pass the threshold -- a trained head
        keeps about that many (the reference's demo output holds 13 meshes) -- and let empty-box removal and the
        class-aware NMS thin them out as they do there.  Returns the shift (reported in `config`)."""
        import warnings
        torch = self.torch
        pc = self.scenes[min(self.scenes)][None]
        with torch.no_grad():
            end_points, _ = self.net.detect(pc)
            d = (end_points['objectness_scores'][0, :, 1] - end_points['objectness_scores'][0, :, 0]).double()
            d, _ = torch.sort(d, descending=True)
            keep = max(1, min(int(keep), d.numel() - 1))
            thr = self.net.cfg.config['generation']['dump_threshold']
            logit_thr = float(torch.log(torch.tensor(thr / (1.0 - thr), dtype=torch.float64)))
            shift = logit_thr - float(d[keep - 1] + d[keep]) / 2.0
            self.net.detection.conv3.bias[1] += shift
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")          # the placeholder-size warning: said once in `config`
                ids = self.net.select_proposals(self.net.detect(pc)[0], 'nms', pc)
        self.torch.cuda.synchronize()
        return shift, int(ids.shape[1])

    def sync(self):
        self.torch.cuda.synchronize()

    def worker_begin(self, w):
        self.torch.cuda.set_device(self.device)
        ctx = self.torch.cuda.stream(self.streams[w])
        ctx.__enter__()
        return ctx

    def worker_end(self, w, ctx):
        self.sinks[w].drain()
        self.streams[w].synchronize()
        ctx.__exit__(None, None, None)

    def run_pass(self, w, ids):
        """One forward pass over the scenes `ids` (ISCNet.generate(selection='all') stage by
        stage, network.py) -> (meshes, vertices, triangles, query points)."""
        torch = self.torch
        net, sink = self.nets[w], self.sinks[w]
        pc = torch.stack([self.scenes[i % self.pool] for i in ids])
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)] if self.record_scenes else None
        if ev:
            ev[0].record()
        with torch.no_grad():
            end_points, proposal_features = net.detect(pc)
            if ev:
                ev[4].record()
            sel = net.select_proposals(end_points, self.args.selection, pc)
            if ev:
                ev[1].record()
            if sel.shape[1] == 0:                   # nothing survived the selection (ISCNet.generate returns [] too)
                # the status word is read HERE as well: an FPS abort / range flag raised by detect() belongs to this
                # scene, not to the next one on this worker's stream
                self._lib.raise_status(self._lib.stream_status_bits())
                sink.start_pending()
                if ev:
                    ev[2].record()
                    ev[3].record()
                    self.scene_records.append({"scenes": [int(i) for i in ids], "worker": w, "events": ev, "queries": 0,
                                               "rounds": 0, "meshes": 0, "vertices": 0, "triangles": 0, "failed": False})
                return 0, 0, 0, 0
            gen = net.completion.generator
            # the previous scene's PCIe copy rides behind one decode launch (--blit-round; default the first, longest)
            tm, blit_round = self.timers[0], self.args.blit_round

            def hook(r, depth):
                tm.local.round = r
                if r == blit_round:
                    sink.start_pending()
            gen.round_hook = hook
            # skip propagation -> codes -> completion + the stream's status word (FPS time-out -> raises; an
            # f16-range flag -> one re-run of the stage at the fallback activation scale, raises only if that
            # overflows too)
            meshes = net.reconstruct(end_points, proposal_features, sel, pc,
                                     hook=(lambda codes, cls: ev[2].record()) if ev else None)
            sink.start_pending()          # (dense grid: no rounds; or fewer rounds than --blit-round)
        v, f, _, _ = gen.last_buffers                              # all K meshes: one vertex / one face buffer
        sink.push(v, f)
        if self.args.blit_round < 0:
            sink.start_pending()
        if self.timers[0].enabled:
            with DECODE_LOCK:
                for r, n in enumerate(gen.stats.get('per_round', [])):
                    self.round_points[r] = self.round_points.get(r, 0) + int(n)
        if ev:
            ev[3].record()
            self.scene_records.append({"scenes": [int(i) for i in ids], "worker": w, "events": ev,
                                       "queries": int(gen.stats.get('n_queries', 0)), "rounds": int(gen.stats.get('rounds', 0)),
                                       "meshes": len(meshes), "vertices": int(v.shape[0]), "triangles": int(f.shape[0]),
                                       "failed": False})
        return len(meshes), int(v.shape[0]), int(f.shape[0]), gen.stats.get('n_queries', 0)

    def scene_stats(self):
        """per-scene records of the timed region with the event times resolved (call after the final sync)"""
        out = []
        for r in self.scene_records:
            if r.get("failed"):
                out.append(r)
                continue
            e = r.pop("events")
            r["ms"] = {"backbone_voting_proposal": e[0].elapsed_time(e[4]), "proposal_selection": e[4].elapsed_time(e[1]),
                       "skip_propagation": e[1].elapsed_time(e[2]),
                       "completion_mise_decoder_marching_cubes": e[2].elapsed_time(e[3]), "total": e[0].elapsed_time(e[3])}
            out.append(r)
        return out

    def decode_totals(self):
        tot = [0.0, 0, 0]
        for tm in self.timers:
            a = tm.totals()
            tot = [x + y for x, y in zip(tot, a)]
        return tot

    def per_round(self):
        """the decoder's launches of the timed region by MISE round (generator.py:99-117): launches, real query points,
        HIP-event ms, algorithmic TFLOP/s -- locates where the kernel is slower inside a scene than alone"""
        agg = {}
        for tm in self.timers:
            for r, a in tm.by_round().items():
                b = agg.setdefault(r, [0.0, 0, 0])
                for i in range(3):
                    b[i] += a[i]
        rows = []
        for r in sorted(agg):
            ms, padded, launches = agg[r]
            real = self.round_points.get(r, padded)
            tf = real * FLOP_PER_QUERY / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            rows.append({"round": r, "launches": launches, "real_points": int(real), "padded_points": int(padded),
                         "ms": ms, "avg_launch_ms": ms / max(launches, 1), "achieved": tf, "frac": tf / MFMA_PEAK_TFLOPS})
        return rows

    def set_timing(self, on):
        for tm in self.timers:
            tm.enabled = on
            if on:
                tm.records = []
        if on:
            self.round_points = {}
        self.record_scenes = bool(on and (getattr(self.args, "stats_out", None) or self.args.selection == "nms"))
        if on:
            self.scene_records = []

    def final_check(self):
        self._lib.device_status()
        self.selfcheck = self.decoder_selfcheck(self.dec if hasattr(self, "dec") else self.nets[0].completion.decoder)
        self.alone = self.decoder_alone()

    def decoder_alone(self):
        """After the timed region, nothing else on the GPU: MISE round 0 of one scene (every proposal's level-0 lattice,
        the shared query list) decoded three times -- the same kernel, shape and launch path as the timed region's round-0
        launches, so `roofline.per_round[0]` can be read against it (inside a scene the launch shares the chip with the
        other scenes' kernels; its event-bracketed time is the kernel's own plus what it lent out)."""
        torch = self.torch
        if not getattr(self, "nets", None):
            return None                       # the stress configuration IS the decoder alone
        net = self.nets[0]
        gen = net.completion.generator
        if gen.upsampling_steps == 0:
            return None
        dec = net.completion.decoder
        K = 256
        g = torch.Generator(device=self.device).manual_seed(7)
        with torch.no_grad():
            c = torch.randn(K, dec.blocks[0].bn_0.c_dim, device=self.device, generator=g)
            table, fcp = dec.fold(torch.zeros(K, dec.z_dim, device=self.device), c)
            pts, lin, tile_prop, tile_src, total = gen._round0(gen.resolution0, gen.upsampling_steps, 1 + gen.padding, K,
                                                               self.device)
            orig = self.timers[0]._orig
            orig(pts, tile_prop, table, fcp, tile_src=tile_src)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                orig(pts, tile_prop, table, fcp, tile_src=tile_src)
            e1.record()
            e1.synchronize()
        ms = e0.elapsed_time(e1) / 3
        tf = total * FLOP_PER_QUERY / (ms * 1e-3) / 1e12
        return {"what": "MISE round 0 of one scene (256 proposals x %d lattice points) decoded alone after the timed region"
                        % (total // K), "avg_launch_ms": ms, "achieved": tf, "frac": tf / MFMA_PEAK_TFLOPS}

    def decoder_selfcheck(self, dec):
        """After the timed region: the same multi-proposal ragged launch four times -- the results must be
        bit-identical (a race shows up as one wave's 16 points differing in one of the runs; the static-priority
        build of round 2 failed exactly this check) -- AND agree with the independent four-wave kernel
        (csrc/occ_decoder.hip: other MFMA shape, other fragment order, no shared scheduling) to a few ulps, so a
        deterministic error of the shipped kernel cannot hide behind its own repeatability."""
        torch = self.torch
        g = torch.Generator(device=self.device).manual_seed(5)
        K, T = 8, 1024
        p = (torch.rand(K, T, 3, device=self.device, generator=g) - 0.5) * 1.1
        c = torch.randn(K, 512, device=self.device, generator=g)
        z = torch.zeros(K, 32, device=self.device)
        orig = self.timers[0]._orig if hasattr(self.timers[0], "_orig") else dec.decode_tiles
        with torch.no_grad():
            table, fcp = dec.fold(z, c)
            tile_prop = torch.arange(K, dtype=torch.int32, device=self.device).repeat_interleave(T // 128)
            pts = p.reshape(-1, 3).contiguous()
            runs = [orig(pts, tile_prop, table, fcp) for _ in range(4)]
            kern = dec.kernel
            other = "w4" if kern == "w8" else "w8"
            try:
                dec.kernel = other
                ref = orig(pts, tile_prop, table, fcp)
            finally:
                dec.kernel = kern
        same = all(torch.equal(runs[0], r) for r in runs[1:])
        if not same:
            raise RuntimeError("decoder self-check: repeated launches on the same input differ")
        d = float((runs[0] - ref).abs().max())
        if not d < 5e-6:
            raise RuntimeError("decoder self-check: %s and %s kernels differ by %.3g" % (kern, other, d))
        return "4 launches of 8 x 1024 points bit-identical; max |%s - %s kernel| = %.2e" % (kern, other, d)

    def kernel_name(self):
        return "%s<%d>" % ("occ_decode8_kernel" if self.nets[0].completion.decoder.kernel == "w8"
                           else "occ_decode_kernel", 3 if self.args.mode == "f16x3" else 1)

    def parity_sample(self, picks=(0, 97, 255)):
        """After the timed region, rank 0, N=1 (beside the CPU baseline): BASELINE.json configs[4]'s parity figure
        on a few proposals of scene 0 -- the CPU path (oracle decoder -> oracle octree MISE -> oracle marching
        cubes; oracle/parity.py, checker only) against this run's HIP value grids and meshes: occupancy IoU of
        `grid >= threshold`, face counts, vertex Hausdorff distance in cells."""
        import numpy as np
        from collections import OrderedDict
        from oracle import oracle, parity
        torch = self.torch
        net = self.nets[0]
        gen = net.completion.generator
        if gen.upsampling_steps == 0:
            return None
        pc = self.scenes[0][None]
        with torch.no_grad():
            end_points, feats = net.detect(pc)
            sel = net.select_proposals(end_points, 'all', pc)
            codes = net.object_codes(end_points, feats, sel, pc)
            cls = net.cls_codes(end_points, sel)
            grids = gen.generate_grids(codes, cls)
            thr = gen.logit_threshold()
            inside = (grids >= thr).flatten(1).float().mean(1)
            thin = int(torch.where(inside > 0, inside, torch.ones_like(inside)).argmin().item())
            picks = list(dict.fromkeys(list(picks) + [thin]))
            idx = torch.tensor(picks, device=codes.device)
            meshes = gen.extract_meshes(grids[idx])
        model = net.completion
        blob = oracle.decoder_param_blob(OrderedDict((k, v.detach().cpu().numpy())
                                                     for k, v in model.decoder.state_dict().items()))
        c_in = codes[idx]
        if getattr(model, 'use_cls_for_completion', False):
            c_in = torch.cat([c_in, cls[idx]], dim=-1)
        z = model.get_z_from_prior((len(picks),), sample=gen.sample, device=codes.device).cpu().numpy()
        c_np = c_in.cpu().numpy()
        rows = []
        for j, k in enumerate(picks):
            cpu_grid, n_q = parity.cpu_value_grid(blob, z[j], c_np[j], gen.resolution0, gen.upsampling_steps, thr,
                                                  gen.padding)
            r = parity.compare(grids[k].cpu().numpy(), meshes[j].vertices.cpu().numpy(),
                               meshes[j].faces.cpu().numpy(), cpu_grid, thr, gen.padding)
            r["proposal"], r["cpu_queries"] = int(k), int(n_q)
            rows.append(r)
            self.cpu_grids = getattr(self, "cpu_grids", []) + [cpu_grid]     # the CPU baseline's octree / MC legs reuse them
        self.cpu_threshold = thr
        return {"what": "CPU path (oracle decoder + octree MISE + marching cubes) vs HIP path, scene 0, proposals %s"
                        % picks,
                "min_iou": min(r["iou"] for r in rows), "flips": sum(r["flips"] for r in rows),
                "max_abs_dlogit": max(r["max_abs_dlogit"] for r in rows),
                "faces_equal": all(r["faces_hip"] == r["faces_cpu"] for r in rows),
                "max_vertex_hausdorff_cells": max(r["hausdorff_cells"] for r in rows),
                "max_iso_residual_logit": max(r["iso_residual_logit"] for r in rows), "per_proposal": rows}


class StressBackend(HipBackend):
    """configs[2]: the decoder alone on 256 proposals x 262 144 uniform query points in the padded unit
    cube, codes ~ N(0,1).  A step = one decode of all 67.1 M points."""

    name = "stress"
    K, T = 256, 262144

    def __init__(self, args, rank, local_rank, world):
        import numpy as np
        import torch
        from rfdnet_amd import _lib, synthetic
        from rfdnet_amd.iscnet import occ_decoder
        assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
        self.one_dev = os.environ.get("RFD_BENCH_ONE_DEVICE") == "1"
        dev_index = 0 if self.one_dev else local_rank
        torch.cuda.set_device(dev_index)
        self.device = torch.device("cuda", dev_index)
        self.dist_backend = "gloo" if self.one_dev else "nccl"
        self.args, self.torch, self._lib = args, torch, _lib
        self.S, self.NB, self.world, self.pool = 1, 1, world, world
        self.round_points = {}
        dec = occ_decoder.DecoderCBatchNorm(dim=3, z_dim=32, c_dim=512, hidden_size=256)
        synthetic.load_seeded(dec, 3)
        dec = dec.to(self.device).eval()
        dec.mode = occ_decoder.MODE_F16X3 if args.mode == "f16x3" else occ_decoder.MODE_F16X1
        self.dec = dec
        self.timers = [DecodeTimer(dec)]
        self.streams = [torch.cuda.Stream(self.device)]
        rng = np.random.default_rng(100 + rank)
        K, T = self.K, self.T
        self.pts = torch.from_numpy(((rng.random((K * T, 3)) - 0.5) * 1.1).astype(np.float32)).to(self.device)
        c = torch.from_numpy(rng.normal(0, 1, (K, 512)).astype(np.float32)).to(self.device)
        z = torch.zeros(K, 32, device=self.device)
        with torch.no_grad():
            self.table, self.fc_p_w = dec.fold(z, c)
        self.tile_prop = torch.arange(K, dtype=torch.int32, device=self.device).repeat_interleave(T // 128)
        torch.cuda.synchronize()

    def worker_end(self, w, ctx):
        self.streams[w].synchronize()
        ctx.__exit__(None, None, None)

    def run_pass(self, w, ids):
        with self.torch.no_grad():
            for _ in ids:
                self.dec.decode_tiles(self.pts, self.tile_prop, self.table, self.fc_p_w)
        return self.K * len(ids), 0, 0, self.K * self.T * len(ids)

    def final_check(self):
        self._lib.device_status()
        self.selfcheck = self.decoder_selfcheck(self.dec)
        self.alone = None
        self.modes = self.compare_modes()

    def compare_modes(self):
        """configs[2] names "bf16 MFMA MLP": the single-pass 16-bit mode of the same kernel (occ_decode8_kernel<1>: one
        f16 MFMA per product instead of three on (hi, lo) splits; f16 carries 3 more significand bits than bf16 at the
        same matrix-core rate, so it bounds what a bf16 build could deliver in accuracy from above and equals it in
        speed) next to the parity mode, on the SAME launch: both timed back to back after the timed region (HIP events on
        the launch stream, 3 launches each after one warm-up), and their logits compared point by point."""
        torch = self.torch
        from rfdnet_amd.iscnet import occ_decoder
        orig = self.timers[0]._orig
        out = {}
        logits = {}
        with torch.no_grad():
            for name, mode in (("f16x3", occ_decoder.MODE_F16X3), ("f16x1", occ_decoder.MODE_F16X1)):
                logits[name] = orig(self.pts, self.tile_prop, self.table, self.fc_p_w, mode=mode)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    orig(self.pts, self.tile_prop, self.table, self.fc_p_w, mode=mode)
                e1.record()
                e1.synchronize()
                ms = e0.elapsed_time(e1) / 3
                tf = self.K * self.T * FLOP_PER_QUERY / (ms * 1e-3) / 1e12
                out[name] = {"kernel": "occ_decode8_kernel<%d>" % mode, "avg_launch_ms": ms, "achieved": tf,
                             "frac": tf / MFMA_PEAK_TFLOPS, "points_per_s": self.K * self.T / (ms * 1e-3)}
            st = self._lib.stream_status_bits()
            d = (logits["f16x1"] - logits["f16x3"]).abs()
            out["f16x1"].update({
                "max_abs_dlogit_vs_f16x3": float(d.max()), "mean_abs_dlogit_vs_f16x3": float(d.mean()),
                "sign_agreement_vs_f16x3": float(((logits["f16x1"] >= 0) == (logits["f16x3"] >= 0)).double().mean()),
                "points_compared": int(d.numel()), "status_bits": int(st),
                "note": "throughput mode, NOT parity (north_star: 1e-4 on logits); one 16-bit MFMA per product, f32 "
                        "accumulate; sign = inside / outside at the 0.5 occupancy threshold"})
        return out

    def kernel_name(self):
        return "%s<%d>" % ("occ_decode8_kernel" if self.dec.kernel == "w8" else "occ_decode_kernel",
                           3 if self.args.mode == "f16x3" else 1)


class StubBackend(object):
    """RFD_BENCH_STUB=1: the launcher / sharding / statistics path with a CPU stand-in for the
    scene (tests/test_bench_launcher.py; never a measurement).  Scene ids listed in
    RFD_BENCH_STUB_FAIL raise, to exercise the per-scene failure accounting."""

    name = "stub"
    dist_backend = "gloo"

    def __init__(self, args, rank, local_rank, world):
        import torch
        self.torch = torch
        self.device = torch.device("cpu")
        self.args, self.S, self.NB, self.world = args, max(1, args.in_flight), max(1, args.batch), world
        self.fail = {int(x) for x in os.environ.get("RFD_BENCH_STUB_FAIL", "").split(",") if x}
        self.seen = []
        self.record_scenes, self.scene_records = bool(getattr(args, "stats_out", None)), []
        self.slow = float(os.environ.get("RFD_BENCH_STUB_SLOW_RANK", "-1")) == rank

    def sync(self):
        pass

    def worker_begin(self, w):
        return None

    def worker_end(self, w, ctx):
        pass

    def run_pass(self, w, ids):
        time.sleep(0.02 if self.slow else 0.002)
        self.seen += list(ids)
        if self.fail & set(ids):
            raise RuntimeError("stub scene %s failed" % sorted(self.fail & set(ids)))
        if self.record_scenes:
            self.scene_records.append({"scenes": [int(i) for i in ids], "worker": w, "failed": False,
                                       "queries": sum(100 + i for i in ids), "ms": {"total": 2.0}})
        return 256 * len(ids), 1000 * len(ids), 2000 * len(ids), sum(100 + i for i in ids)

    def scene_stats(self):
        return list(self.scene_records)

    def decode_totals(self):
        return [1.0, 128, 1]

    def set_timing(self, on):
        if on:
            self.scene_records = []

    def final_check(self):
        pass

    def kernel_name(self):
        return "stub"


# ------------------------------------------------------------------------ job ---
def run_job(args, be, rank, world, dist):
    """warm-up, barrier, K timed steps, barrier -> per-rank statistics vector."""
    from concurrent.futures import ThreadPoolExecutor
    S, NB = be.S, be.NB
    per_step = S * NB * world                              # scenes per step over the whole job

    def shard(first_step, n_steps):
        """this rank's scenes of steps [first_step, first_step + n_steps), split over its workers"""
        lo, hi = first_step * per_step, (first_step + n_steps) * per_step
        mine = [i for i in sharding.scene_ids_for_rank(hi, rank, world) if i >= lo]
        return [sharding.scene_ids_for_worker(mine, w, S, NB) for w in range(S)]

    def worker(w, passes):
        tot, failed = [0, 0, 0, 0], 0
        ctx = be.worker_begin(w)
        try:
            for ids in passes:
                try:
                    try:
                        r = be.run_pass(w, ids)
                    except Exception as e:
                        # status bit 0 = a multi-workgroup FPS launch aborted because its workgroups were not resident
                        # together for 500 ms (a transient co-residency stall on a shared GPU): the stream and its
                        # exchange region are good again, so the scene gets ONE more try before it counts as failed
                        if not getattr(e, "status", 0) & 1:
                            raise
                        sys.stderr.write("[rank %d] scene(s) %s: FPS abort, retrying once\n" % (rank, ids))
                        be.retried = getattr(be, "retried", 0) + len(ids)
                        r = be.run_pass(w, ids)
                    tot = [a + b for a, b in zip(tot, r)]
                except Exception as e:                    # scene marked failed, the sweep goes on
                    failed += len(ids)
                    if getattr(be, "record_scenes", False):
                        be.scene_records.append({"scenes": [int(i) for i in ids], "worker": w, "failed": True,
                                                 "error": "%s: %s" % (type(e).__name__, e)})
                    sys.stderr.write("[rank %d] scene(s) %s failed: %s: %s\n" % (rank, ids, type(e).__name__, e))
        finally:
            be.worker_end(w, ctx)
        return tot, failed

    pool = ThreadPoolExecutor(max_workers=S)
    if args.warmup:
        list(pool.map(lambda w: worker(w, shard(0, args.warmup)[w]), range(S)))
    be.sync()
    if dist is not None:
        dist.barrier()
    be.sync()
    be.set_timing(True)
    plan = shard(args.warmup, args.steps)
    if getattr(args, "scenes", None):
        # a sweep of exactly N scenes: global scene j of the timed region (id = warm-up scenes + j) -> rank j mod world,
        # the partition of sharding.scene_ids_for_rank (tests/test_bench_launcher.py: 311 scans -> 39 x 7 + 38)
        first = args.warmup * per_step
        mine = [first + j for j in sharding.scene_ids_for_rank(args.scenes, rank, world)]
        plan = [sharding.scene_ids_for_worker(mine, w, S, NB) for w in range(S)]
    t0 = time.perf_counter()
    cpu0 = time.process_time()                               # CPU seconds of every thread of this rank
    res = list(pool.map(lambda w: worker(w, plan[w]), range(S)))
    be.sync()
    be.host_cpu_s = time.process_time() - cpu0               # (before the barrier: a waiting rank may spin in it)
    if dist is not None:
        dist.barrier()
    be.sync()
    elapsed = time.perf_counter() - t0
    dec_ms, dec_pts, dec_launches = be.decode_totals()
    be.last_scene_stats = be.scene_stats() if (getattr(be, "record_scenes", False) and hasattr(be, "scene_stats")) else None
    if getattr(args, "stats_out", None) and be.last_scene_stats is not None:
        path = args.stats_out if rank == 0 else "%s.rank%d" % (args.stats_out, rank)
        with open(path, "w") as fh:
            json.dump({"rank": rank, "world": world, "config": args.config, "timed_region_s": elapsed,
                       "scenes": be.last_scene_stats}, fh, indent=1)
    be.set_timing(False)
    be.final_check()
    n_meshes, nv, nt, nq = (sum(r[0][i] for r in res) for i in range(4))
    failed = sum(r[1] for r in res)
    n_scenes = sum(len(ids) for p in plan for ids in p)
    single = None
    if not args.no_latency and be.name == "hip" and rank == 0:
        # one scene at a time on one stream: the latency reading of "a single scene per GPU"
        L = 3
        worker(0, [[0]])
        be.sync()
        be.record_scenes, be.scene_records = True, []
        t1 = time.perf_counter()
        worker(0, [[i * world] for i in range(L)])
        be.sync()
        single = (time.perf_counter() - t1) / L
        recs = [r for r in be.scene_stats() if not r.get("failed") and "ms" in r]
        be.record_scenes = False
        be.single_stage_ms = {k: sum(r["ms"][k] for r in recs) / len(recs) for k in sorted(recs[0]["ms"])} if recs else None
    stats = sharding.pack_stats(steps=n_scenes - failed, elapsed_s=elapsed, n_meshes=n_meshes, n_vertices=nv,
                                n_triangles=nt, n_queries=nq, decode_ms=dec_ms, decode_points=dec_pts,
                                decode_launches=dec_launches, failed=failed)
    return stats, single


GROUP_TRAFFIC_FILE = os.path.join(ROOT, "profiles", "group_traffic.json")


def grouping_roofline(device):
    """HBM figure of the grouping family (north_star: "rocprof HBM GB/s reported for grouping"), measured
    live with HIP events on the stream the kernel is launched on: QueryAndGroup's fused epilogue
    (rfd_group_concat) at the SA2 layer's shape (131 channels x 1024 centres x 32 samples from a
    2048-point table), B = 32 scenes per launch (one scene, 18.4 MB, is launch-bound) and B = 8
    beside it.  Algorithmic bytes per launch = 4 C M ns written + 4 M ns of indices + the (C, N) table and
    xyz read once.  `traffic` = HBM bytes per launch from the committed PMC passes over the same launches
    (profiles/group_traffic.json; counters cannot be read from inside this process)."""
    import torch
    from rfdnet_amd.pointnet2_ops import _ext
    try:
        with open(GROUP_TRAFFIC_FILE) as fh:
            pmc = json.load(fh)
    except (OSError, ValueError):
        pmc = {}
    N, M, ns, C = 2048, 1024, 32, 128
    out = {}
    for B in (32, 8):
        g = torch.Generator(device=device).manual_seed(0)
        xyz = torch.rand(B, N, 3, device=device, generator=g)
        ctr = xyz[:, :M].contiguous()
        feats = torch.randn(B, C, N, device=device, generator=g)
        idx = _ext.ball_query(ctr, xyz, 0.4, ns)
        nbytes = B * ((3 + C) * M * ns * 4 + M * ns * 4 + C * N * 4 + N * 12 + M * 12)
        for _ in range(3):
            _ext.group_concat(xyz, ctr, feats, idx, 0.4, True, True, False)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        it = 10
        e0.record()
        for _ in range(it):
            _ext.group_concat(xyz, ctr, feats, idx, 0.4, True, True, False)
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / it
        ach = nbytes / ms / 1e6
        t = pmc.get(str(B), {}).get("bytes_per_launch")
        out[B] = {"achieved": ach, "frac": ach / 8000.0, "algorithmic_bytes_per_launch": nbytes, "avg_launch_ms": ms,
                  "traffic": t}
    r = {"bound": "hbm", "kernel": "group_lds_kernel (rfd_group_concat, SA2 shape, 32 scenes per launch)",
         "achieved": out[32]["achieved"], "peak": 8000.0, "unit": "GB/s", "frac": out[32]["frac"],
         "algorithmic_bytes_per_launch": out[32]["algorithmic_bytes_per_launch"], "avg_launch_ms": out[32]["avg_launch_ms"],
         "traffic": out[32]["traffic"], "traffic_source": pmc.get("source"),
         "b8": out[8],
         "note": "not on the headline path any more (the fused SA layer never materialises the grouped tensor); "
                 "reported because the grouping family is the path's HBM-bound op"}
    return r


VALU_PEAK_TFLOPS = 157.3      # fp32 vector peak, MI355X_MICROARCH.md (256 CUs x 4 SIMD x 64 lanes x 2 flop x 2.4 GHz)


def pointop_rooflines(device, pc):
    """SURVEY 8(d)'s per-op figures for the two point ops that are NOT HBM-bound, measured live (HIP events on the stream
    the kernels are launched on) on the benchmark's own scene, at the call sites' shapes (pointnet2backbone.py:27-61,
    skip_propagation.py:24-31):
      furthest-point sampling (latency / on-chip bound): point-updates/s = N (M-1) / t and effective on-chip GB/s =
        16 B N (M-1) / t -- what a kernel re-streaming the points every round (the reference's, sampling_gpu.cu:69-173)
        would have to move; this kernel keeps them in registers and touches HBM for 12 N + 4 N + 4 M bytes only;
      ball query (VALU / LDS bound): centre-point distance tests/s = M N / t (8 flop each).
    `frac` is against the fp32 VALU peak at 8 flop per update / test; FPS runs on G <= 32 workgroups by design (the
    exchange between workgroups every round is what bounds it), so its frac is a statement about latency, not ALUs."""
    import torch
    from rfdnet_amd.pointnet2_ops import _ext
    xyz = pc[:, :3].contiguous()[None]
    N = int(xyz.shape[1])

    def timed(fn, it):
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(it):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / it

    peak_tests = VALU_PEAK_TFLOPS * 1e12 / 8.0
    M = 2048
    ms = timed(lambda: _ext.furthest_point_sampling(xyz, M), 5)
    upd = N * (M - 1) / (ms * 1e-3)
    fps = {"bound": "latency (inter-workgroup exchange per round); on-chip", "kernel": "fps_kernel, SA1 %d -> %d" % (N, M),
           "avg_launch_ms": ms, "us_per_round": 1e3 * ms / (M - 1), "achieved": upd / 1e9, "unit": "G point-updates/s",
           "peak": peak_tests / 1e9, "frac": upd / peak_tests, "effective_onchip_GBps": 16.0 * upd / 1e9,
           "algorithmic_hbm_bytes_per_launch": 12 * N + 4 * N + 4 * M}
    inds = _ext.furthest_point_sampling(xyz, M)
    ctr = torch.gather(xyz, 1, inds.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    out = {}
    for name, c, r, ns in (("sa1", ctr, 0.2, 64), ("skip_propagation", ctr[:, :256].contiguous(), 1.0, 1024)):
        m = int(c.shape[1])
        ms = timed(lambda: _ext.ball_query(c, xyz, r, ns), 20)
        tests = m * N / (ms * 1e-3)
        out[name] = {"shape": "%d centres x %d points, radius %.1f, nsample %d" % (m, N, r, ns), "avg_launch_ms": ms,
                     "achieved": tests / 1e9, "frac": tests / peak_tests,
                     "algorithmic_hbm_bytes_per_launch": 12 * N + 12 * m + 4 * m * ns}
    bq = {"bound": "valu/lds", "kernel": "ball_query_kernel", "unit": "G distance tests/s", "peak": peak_tests / 1e9,
          "achieved": out["sa1"]["achieved"], "frac": out["sa1"]["frac"], "avg_launch_ms": out["sa1"]["avg_launch_ms"],
          "sa1": out["sa1"], "skip_propagation": out["skip_propagation"],
          "note": "upper bound on the work: a workgroup stops scanning once all its centres hold nsample hits"}
    return fps, bq


def encoder_gemm_roofline():
    """The second-largest kernel of a scene, measured live like the two point ops: the ten GEMMs of the skip-propagation
    encoder (ResnetPointnet, layers.py:340-392: 256 proposals x 1024 points = 262 144 rows; fc_0 512 x 1024|512, [fc_1 |
    shortcut] 512 x 1536|1024; the last block's second GEMM only pools) on fragment-ordered split activations, at the
    headline shapes with synthetic operands, HIP events on the launch stream.  `achieved` = ALGORITHMIC flop (2 M N K per
    GEMM; the f16x3 scheme issues 3x that on the matrix pipe) over the ten launches' time."""
    import torch
    from rfdnet_amd import gemm
    M, h, T = 262144, 512, 1024
    sa = gemm.SA
    g = torch.Generator(device="cuda").manual_seed(0)

    def timed(fn, it=5):
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(it):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / it

    ms = flop = 0.0
    per = {}
    hb = h // 32
    gb = torch.randn(M // T, h, device="cuda", generator=g)
    pool = torch.zeros(M // T, h, device="cuda")
    for name, kin, n_full, n_pool in (("block 0", 2 * h, 1, 0), ("blocks 1-4", h, 3, 1)):
        w1 = torch.randn(h, kin, device="cuda", generator=g) * 0.05
        w2 = torch.randn(h, h + kin, device="cuda", generator=g) * 0.05
        rows = torch.randn(M, h + kin, device="cuda", generator=g)
        fcat = gemm.rows_to_frag(rows, sa=sa)
        fnxt = gemm.frag_empty(M, 2 * h, "cuda")
        t1 = timed(lambda: gemm.linear_frag(fcat[:, hb:], w1, gbias=gb, rows_per_group=T, out=fcat[:, :hb], sa=sa))
        fcat = gemm.rows_to_frag(rows, sa=sa)                           # fc_0 wrote into the hidden window: fresh operands for the second GEMM
        del rows
        t2 = timed(lambda: gemm.linear_frag(fcat, w2, gbias=gb, rows_per_group=T, out=fnxt[:, hb:], pool=pool, sa=sa))
        t2p = timed(lambda: gemm.linear_frag(fcat, w2, gbias=gb, rows_per_group=T, pool=pool, store=False, sa=sa)) if n_pool else 0.0
        fl1, fl2 = 2.0 * M * h * kin, 2.0 * M * h * (h + kin)
        ms += (n_full + n_pool) * t1 + n_full * t2 + n_pool * t2p
        flop += (n_full + n_pool) * fl1 + (n_full + n_pool) * fl2
        per[name] = {"fc_0_ms": t1, "fc_1_shortcut_ms": t2, "fc_1_shortcut_pool_only_ms": t2p or None,
                     "fc_0_tflops": fl1 / t1 / 1e9, "fc_1_shortcut_tflops": fl2 / t2 / 1e9}
        del fcat, fnxt
    _ = torch.cuda.current_stream().synchronize()
    hip_status = None
    try:
        from rfdnet_amd import _lib
        hip_status = int(_lib.stream_status_bits())       # synthetic operands may trip the f16-range flag: read and clear it
    except Exception:
        pass
    ach = flop / (ms * 1e-3) / 1e12
    return {"bound": "mfma", "kernel": "gemm_rowsf_kernel (csrc/gemm_f16x3.hip), ten launches of one scene's encoder",
            "ms_per_scene": ms, "achieved": ach, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / MFMA_PEAK_TFLOPS,
            "algorithmic_flop_per_scene": flop, "blocks": per, "status_bits_cleared": hip_status,
            "note": "f16x3 split GEMMs: three matrix instructions per product, so 1/3 of the dense f16 peak is this scheme's ceiling"}


def mfma_busy(mode):
    """matrix-pipe busy share of the decoder's SIMD time from the committed counter passes (rocprofv3 --pmc
    SQ_VALU_MFMA_BUSY_CYCLES / SQ_WAVE_CYCLES over tools/dec_only.py; counters cannot be read from inside this
    process) -> (fraction or None, source)."""
    try:
        with open(MFMA_BUSY_FILE) as f:
            d = json.load(f)
        return d.get(mode, {}).get("mfma_busy"), d.get("source")
    except (OSError, ValueError):
        return None, None


def traffic_per_query():
    """HBM bytes per query point of the decoder from the committed PMC passes (rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE in separate runs of this benchmark; counters cannot be read from
    inside the process).  None when the file is absent."""
    try:
        with open(TRAFFIC_FILE) as f:
            d = json.load(f)
        return float(d["bytes_per_query"]), d.get("source", TRAFFIC_FILE)
    except (OSError, ValueError, KeyError):
        return None, None


# free HBM a rank needs before it starts, by config (GiB): peak reserved memory of a run with the default scenes in
# flight (`config.hbm_peak_gib` of the bench line), rounded up with ~50 % head room
HBM_NEED_GIB = {"headline": 24.0, "mise128": 48.0, "stress": 8.0, "dense32": 12.0, "demo": 4.0}


def main(argv=None):
    args = parse(argv)
    if args.gpus > 1 and not sharding.launched():
        # `python bench.py --gpus N`: become the launcher of N ranks (one per GPU)
        sys.exit(sharding.launch_local_ranks(os.path.abspath(__file__), sys.argv[1:] if argv is None else argv,
                                             args.gpus))
    rank, local_rank, world = sharding.rank_env()
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d; launch with --nproc-per-node %d (or unset WORLD_SIZE "
                 "and let bench.py spawn its own ranks)" % (args.gpus, world, args.gpus))
    cls = StubBackend if os.environ.get("RFD_BENCH_STUB") == "1" else \
        StressBackend if args.config == "stress" else HipBackend
    if args.preflight:
        allowed0 = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else -1
        if cls is not StubBackend and os.environ.get("RFD_PIN_NUMA", "1") != "0" and not os.environ.get("RFD_BENCH_ONE_DEVICE"):
            cpus = sharding.pin_cpus_for_rank(local_rank)
            if cpus:
                sharding._pin(cpus)                     # report the affinity the job itself would run with
        g = sharding.preflight(rank, local_rank, world, stub=cls is StubBackend,
                               one_device=os.environ.get("RFD_BENCH_ONE_DEVICE") == "1", allowed_cpus=allowed0,
                               need_gib=HBM_NEED_GIB.get(args.config, 8.0))
        if rank == 0:
            print(json.dumps({"preflight": "ok", "n_gpus": world,
                              "ranks": [dict(zip(sharding.PREFLIGHT_FIELDS, [float(x) for x in row])) for row in g]}))
            sys.stdout.flush()
        return
    all_cpus = None
    if cls is not StubBackend and os.environ.get("RFD_PIN_NUMA", "1") != "0" and not os.environ.get("RFD_BENCH_ONE_DEVICE"):
        # host threads, pinned mesh buffers (first touch) and the HIP runtime's helper threads on the CPUs of THIS GPU's
        # NUMA node -- also with one rank, and when torch.distributed.run (not sharding.launch_local_ranks) started the
        # ranks.  +1-2 % scenes/s on a two-socket box (profiles/r04_numa_prime.txt).
        cpus = sharding.pin_cpus_for_rank(local_rank)
        if cpus:
            all_cpus = os.sched_getaffinity(0)
            sharding._pin(cpus)
    be = cls(args, rank, local_rank, world)
    if all_cpus and cls is not StubBackend:
        # the pinning above was chosen BEFORE the runtime came up, from the KFD topology; now HIP says which PCI device
        # this rank really drives: if its NUMA node is another one, re-pin (threads created from here on follow)
        want = sharding.numa_cpus_for_bdf(sharding.device_bdf(be.torch.cuda.get_device_properties(be.device)) or "")
        if want and not set(os.sched_getaffinity(0)) <= set(want):
            want = sorted(set(want) & set(all_cpus))
            if want:
                sys.stderr.write("[rank %d] re-pinning to the NUMA node of the GPU's PCI address (%d CPUs)\n" % (rank, len(want)))
                sharding._pin(want)
    dist = None
    # RFD_BENCH_FORCE_DIST=1: initialise the process group even for one rank (exercises the RCCL barrier /
    # all-gather on a 1-GPU box: `python -m torch.distributed.run --nproc-per-node 1 bench.py`)
    if world > 1 or (os.environ.get("RFD_BENCH_FORCE_DIST") == "1" and sharding.launched()):
        import torch.distributed as dist
        kw = {}
        if be.dist_backend == "nccl":
            kw["device_id"] = be.device          # bind the communicator to this rank's GPU up front
        dist.init_process_group(backend=be.dist_backend, **kw)
    stats, single = run_job(args, be, rank, world, dist)
    gathered = sharding.gather_stats(stats, be.device, dist)     # the path's only exchange step
    value_all, t_max = sharding.job_throughput(gathered)

    if rank == 0:
        F = sharding.STAT_FIELDS.index
        scenes_total = float(gathered[:, F("steps")].sum())
        dec_ms = float(gathered[:, F("decode_ms")].sum())
        dec_pts = float(gathered[:, F("n_queries")].sum())      # real query points (tile padding excluded)
        dec_launches = float(gathered[:, F("decode_launches")].sum())
        failed = int(gathered[:, F("failed")].sum())
        ach = dec_pts * FLOP_PER_QUERY / (dec_ms * 1e-3) / 1e12 if dec_ms > 0 else 0.0
        bpq, bpq_src = traffic_per_query()
        cfgd = dict(CONFIGS[args.config], points=args.points, resolution0=args.resolution0,
                    upsampling_steps=args.upsampling_steps, res=args.resolution0 << args.upsampling_steps)
        stress = args.config == "stress"
        if stress:
            metric, unit, value = "occupancy-decoder query points/sec (256 proposals x 262144 points)", "points/s", \
                dec_pts / t_max
        else:
            metric, unit, value = "scenes/sec end-to-end reconstruction (ScanNet, %d^3 %s)" % (
                cfgd["res"], "MISE" if args.upsampling_steps else "dense grid"), "scenes/s", value_all
            if args.config == "demo":
                metric = "scenes/sec, the reference's demo workload (objectness + NMS selection, dense 32^3 grids)"
        per = max(scenes_total, 1.0)
        out = {
            "metric": metric, "value": value, "unit": unit, "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * t_max / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (point ops, MLPs); f16x3 split MFMA with f32 accumulate (decoder, fp32-class)"
                     if args.mode == "f16x3" else "f32; f16 MFMA decoder (throughput mode, NOT parity)",
            "data": "synthetic (seeded ScanNet-like scenes, seeded random-init weights)",
            "config": {"workload": cfgd["name"] % cfgd,
                       "proposals_per_scene": int(gathered[:, F("n_meshes")].sum() / per),
                       "queries_per_scene": int(dec_pts / per),
                       "vertices_per_scene": int(gathered[:, F("n_vertices")].sum() / per),
                       "hbm_peak_gib": round(be.torch.cuda.max_memory_reserved() / 2.0 ** 30, 2)
                       if be.name != "stub" else None,
                       # host CPU seconds rank 0 spent per scene inside the timed region (all its threads): what
                       # 8 ranks x `scenes_in_flight` threads ask of the node's cores
                       "host_cpu_s_per_scene_rank0": round(getattr(be, "host_cpu_s", 0.0) / max(float(gathered[0, F("steps")]), 1.0), 5),
                       "scenes_in_flight_per_gpu": be.S * be.NB, "scenes_per_forward": be.NB,
                       "scenes_per_step": be.S * be.NB * world, "scenes_done": int(scenes_total),
                       "scenes_failed": failed, "scenes_retried_after_fps_abort": int(getattr(be, "retried", 0)),
                       "decoder_selfcheck": getattr(be, "selfcheck", None),
                       "parallelism": "scenes sharded across GPUs (scene i -> rank i mod %d), dp%d; %d forward "
                                      "passes of %d scene(s) in flight per GPU" % (world, world, be.S, be.NB)},
            "roofline": {"bound": "mfma", "kernel": be.kernel_name(),
                         "achieved": ach, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": ach / MFMA_PEAK_TFLOPS,
                         "traffic": bpq * dec_pts / dec_launches if (bpq and dec_launches) else None,
                         "traffic_source": ("%.1f B per query point x queries per launch; NOT measured by this "
                                            "process (PMC counters need rocprofv3): %s" % (bpq, bpq_src))
                         if bpq else None,
                         "launches": int(dec_launches),
                         "per_round": be.per_round() if hasattr(be, "per_round") else None,
                         "alone": getattr(be, "alone", None),
                         "avg_launch_ms": dec_ms / dec_launches if dec_launches else None,
                         "algorithmic_flop_per_launch": dec_pts * FLOP_PER_QUERY / dec_launches if dec_launches else None,
                         "note": "algorithmic FLOPs (1 312 768 per query point) over HIP-event time of the decoder "
                                 "launches inside the timed region; f16x3 issues 3x that on the MFMA pipe"},
        }
        if be.name != "stub":
            from rfdnet_amd import _lib as _rfd_lib
            out["roofline"]["tail_route"] = {
                "max_tiles": int(_rfd_lib.lib().rfd_occ_set_tail_tiles(-1)),
                "note": "decoder launches of at most max_tiles tiles (the last MISE rounds) run on occ_decode_tail_kernel "
                        "(csrc/occ_decoder_tail.hip: one wave per 16 points, no LDS, logits bit-identical); they are inside "
                        "`achieved` and `per_round` like every other decoder launch"}
        busy, busy_src = mfma_busy(args.mode)
        out["roofline"]["mfma_busy"] = busy
        out["roofline"]["mfma_busy_source"] = busy_src
        if args.scenes:
            out["config"]["sweep_scenes"] = args.scenes
        modes = getattr(be, "modes", None)
        if modes:
            # configs[2]'s named quantities side by side: the parity mode (the headline of this line), the single-pass
            # 16-bit mode, and the matrix-pipe-busy counter of each
            for m in ("f16x3", "f16x1"):
                modes[m]["mfma_busy"], _ = mfma_busy(m)
            out["roofline"]["f16x3_back_to_back"] = modes["f16x3"]
            out["roofline"]["f16x1"] = modes["f16x1"]
        if args.config == "demo" and be.name == "hip":
            shift, k0 = be.demo_calibration or (None, None)
            out["config"].update({
                "selection": "objectness > dump_threshold + class-aware 3-D NMS + empty-box removal (demo.py:223-234, "
                             "ap_helper.py:131-264) on the device",
                "objectness_bias_shift": shift, "proposals_kept_scene0": k0,
                "placeholder_mean_sizes": bool(getattr(be, "placeholder_sizes", False)),
                "launch_overhead_note": "one scene = ~375 launches, yet the detection stage is kernel-bound: captured into a HIP "
                                        "graph and replayed it takes 6.734 ms against 6.736 ms eager, bit-equal outputs "
                                        "(profiles/r05_graph_probe.txt: FPS is 5.9 ms of it) -- a graph does not pay"})
            recs = [r for r in (getattr(be, "last_scene_stats", None) or []) if not r.get("failed") and "ms" in r]
            if recs:
                keys = sorted(recs[0]["ms"])
                out["stage_ms_per_scene"] = {k: sum(r["ms"][k] for r in recs) / len(recs) for k in keys}
                out["stage_ms_per_scene"]["note"] = ("HIP-event times on each scene's own stream, %d scenes in flight: "
                                                     "stages of different scenes overlap" % (be.S * be.NB))
        if be.name in ("hip", "stress") and not args.no_extras:
            out["roofline_grouping"] = grouping_roofline(be.device)
        if be.name == "hip" and not args.no_extras:
            out["roofline_fps"], out["roofline_ball_query"] = pointop_rooflines(be.device, be.scenes[min(be.scenes)])
            try:
                out["roofline_encoder_gemm"] = encoder_gemm_roofline()
            except Exception as e:                       # never let an extra take the line down
                out["roofline_encoder_gemm"] = {"error": repr(e)}
        if single is not None:
            out["single_scene"] = {"scenes_in_flight": 1, "ms_per_scene": 1e3 * single,
                                   "scenes_per_s": 1.0 / single,
                                   "stage_ms": getattr(be, "single_stage_ms", None)}
        if world == 1 and not args.no_cpu_baseline and be.name != "stub":
            if all_cpus:
                # the CPU legs (parity checker, CPU baseline) get EVERY host core back: the NUMA pinning above is for the
                # GPU path's host threads; threads created from here on (OpenMP, torch intra-op) inherit the full mask
                try:
                    os.sched_setaffinity(0, all_cpus)
                except OSError:
                    pass
            # the checker's CPU port, timed beside -- in a FRESH interpreter with the full affinity mask and explicit
            # OpenMP settings: workers created in THIS process under the NUMA pinning keep that mask (rounds 4-5: 128
            # workers on one node's cores, MLP legs 10x slower)
            from oracle import cpu_baseline
            if stress:
                out["cpu_baseline"] = cpu_baseline.run_best("decoder", cpus=all_cpus)
            else:
                par = be.parity_sample()         # first: its CPU value grids feed the baseline's octree / MC legs
                if par is not None:
                    out["config"]["parity_iou"] = par["min_iou"]
                    out["parity"] = par
                out["cpu_baseline"] = cpu_baseline.run_best(
                    "scene", cpus=all_cpus, points=args.points, resolution0=args.resolution0,
                    upsampling_steps=args.upsampling_steps, n_queries_per_scene=int(dec_pts / per),
                    n_prop=int(gathered[0, F("n_meshes")] / per), scene_grids=getattr(be, "cpu_grids", None),
                    threshold_logit=getattr(be, "cpu_threshold", 0.0))
        print(json.dumps(out))
        sys.stdout.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
