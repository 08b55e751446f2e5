"""GPU (-m gpu): the bench configurations added in round 5 print what the record says they print.
  --config stress   BASELINE configs[2]'s named quantities side by side in ONE line: the parity mode (headline of the
                    line), the single-pass 16-bit mode (`roofline.f16x1`: TFLOP/s, frac, max |dlogit| and sign agreement
                    against the parity logits of the same launch) and the matrix-pipe-busy figure with its source
  --config demo     the reference's `main.py --mode demo` workload (demo.py:200-276): objectness + NMS selection, a
                    dozen proposals, dense 32^3 -- stage times per scene, calibration and placeholder flags in `config`
  --preflight       on hardware: one rank, every field filled in"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*argv, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), env=env, capture_output=True,
                       text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-500:]
    return json.loads(lines[0])


def test_stress_line_carries_both_arithmetic_modes_and_the_pipe_busy_figure(hip):
    out = _bench("--config", "stress", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-extras")
    r = out["roofline"]
    assert out["unit"] == "points/s" and r["kernel"] == "occ_decode8_kernel<3>" and 0.12 < r["frac"] < 1.0 / 3
    x1, x3 = r["f16x1"], r["f16x3_back_to_back"]
    assert x1["kernel"] == "occ_decode8_kernel<1>" and x1["points_compared"] == 256 * 262144
    assert x1["achieved"] > x3["achieved"] > 300.0                   # one MFMA per product is faster ...
    assert 1e-5 < x1["max_abs_dlogit_vs_f16x3"] < 1e-2               # ... and NOT inside the 1e-4 contract
    assert x1["sign_agreement_vs_f16x3"] > 0.999
    assert abs(x3["frac"] - r["frac"]) < 0.02                        # the timed region's figure and the back-to-back one agree
    assert r["mfma_busy"] and 0.3 < r["mfma_busy"] < 1.0 and r["mfma_busy_source"]
    assert x1["mfma_busy"] and x1["mfma_busy"] < r["mfma_busy"]


def test_headline_line_carries_the_per_kernel_rooflines(hip):
    """the extras of the default configuration: the two point ops, the encoder's GEMMs and the tail route, measured live"""
    out = _bench("--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-latency")
    assert out["config"]["scenes_failed"] == 0 and out["roofline"]["tail_route"]["max_tiles"] == 384
    rounds = out["roofline"]["per_round"]
    assert len(rounds) == 3 and rounds[2]["avg_launch_ms"] < 1.5             # the 64^3 tail round runs on the tail kernel
    g = out["roofline_encoder_gemm"]
    assert "error" not in g, g
    assert 4.0 < g["ms_per_scene"] < 8.0 and 0.12 < g["frac"] < 1.0 / 3 and g["status_bits_cleared"] == 0
    assert 0.5 < out["roofline_fps"]["us_per_round"] < 5.0 and out["roofline_ball_query"]["frac"] > 0.01


def test_demo_workload_line(hip):
    out = _bench("--config", "demo", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extras")
    c = out["config"]
    assert "NOT a BASELINE config" in c["workload"] and c["scenes_failed"] == 0 and c["scenes_done"] == 8
    assert c["placeholder_mean_sizes"] is True                      # no datasets/scannet/scannet_means.npz on the box: flagged
    assert 1 <= c["proposals_kept_scene0"] <= 16 and 1 <= c["proposals_per_scene"] <= 40
    st = out["stage_ms_per_scene"]
    for k in ("backbone_voting_proposal", "proposal_selection", "skip_propagation", "completion_mise_decoder_marching_cubes"):
        assert st[k] > 0.0
    one = out["single_scene"]
    assert one["ms_per_scene"] < 40.0 and one["stage_ms"]["backbone_voting_proposal"] < 15.0
    assert out["value"] > 20.0


def test_preflight_on_hardware(hip):
    out = _bench("--preflight", timeout=120)
    assert out["preflight"] == "ok" and out["n_gpus"] == 1
    r = out["ranks"][0]
    assert r["visible_devices"] >= 1 and r["free_gib"] > 24 and r["library_ok"] == 1.0 and r["hw_queues"] == 16
