"""Repeatability of the host-CPU baseline on the box it is quoted on.

    python tools/cpu_baseline_repeat.py [--runs 3] [--out gpurun_out/r06_cpu_baseline_repeat.txt]

Runs oracle/cpu_baseline.run_isolated (fresh interpreter, full affinity mask, one OpenMP thread per physical core)
`--runs` times with the default wait policy, once with the other one and once on HALF the physical cores (one socket of a
two-socket box), and prints one row per run and leg plus the spread.  CPU only: no GPU context is created.
-> profiles/r06_cpu_baseline_repeat.txt"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cpu_baseline  # noqa: E402  (measurement infrastructure: this tool IS the cpu_baseline leg)

LEGS = ("fps", "ball_query", "group_interp", "mlp_backbone", "mlp_vote_proposal", "skip_propagation_nets", "decoder",
        "mise_octree", "marching_cubes")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=3)
    ap.add_argument("--queries", type=int, default=12108113)     # the headline scene's query count (BENCH_r05)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    cpus = cpu_baseline.widest_affinity()
    phys = len(cpu_baseline.physical_cores(cpus))
    lines = ["host: %d CPUs in the mask, %d physical cores; `lscpu`: %s" % (
        len(cpus), phys, " | ".join(l.strip() for l in os.popen("lscpu").read().splitlines()
                                    if l.split(":")[0].strip() in ("Model name", "Socket(s)", "Core(s) per socket",
                                                                   "Thread(s) per core", "NUMA node(s)")))]
    plans = [("passive", None)] * a.runs + [("active", None), ("passive", max(1, phys // 2))]
    rows = []
    for i, (wait, threads) in enumerate(plans):
        t0 = time.time()
        out = cpu_baseline.run_isolated("scene", cpus=cpus, n_queries_per_scene=a.queries, wait_policy=wait,
                                        threads=threads)
        rows.append((wait, out["cores"], out))
        lines.append("run %d  wait=%s threads=%d  scene_s=%.1f  (%.0f s wall)  decoder torch %d pts/s  C %d pts/s" % (
            i, wait, out["cores"], out["scene_s"], time.time() - t0, out["legs"]["decoder"]["torch_points_per_s"],
            out["legs"]["decoder"]["c_points_per_s"]))
        lines.append("   " + "  ".join("%s=%.3f" % (k, out["stage_s"][k]) for k in LEGS if k in out["stage_s"]))
        lines.append("   gflops: " + "  ".join("%s=%s" % (k, out["legs"][k]["gflops"]) for k in LEGS
                                                if out["legs"].get(k, {}).get("gflops")))
    main_runs = [r[2] for r in rows[:a.runs]]
    lines.append("spread over the %d default runs (max/min):" % a.runs)
    for k in LEGS + ("scene_s",):
        v = [r["scene_s"] if k == "scene_s" else r["stage_s"][k] for r in main_runs]
        lines.append("   %-24s %s  -> %.2fx" % (k, " ".join("%.3f" % x for x in v), max(v) / max(min(v), 1e-9)))
    lines.append("isolation: " + json.dumps(main_runs[0]["isolation"]))
    lines.append("sample: " + main_runs[0]["sample"])
    txt = "\n".join(lines)
    print(txt)
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        with open(a.out, "w") as f:
            f.write(txt + "\n")


if __name__ == "__main__":
    main()
