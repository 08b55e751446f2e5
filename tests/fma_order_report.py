"""How much of the point-op output depends on the assumed nvcc contraction of a*a + b*b + c*c?

The reference's CUDA kernels are compiled with nvcc's default -fmad=true; no CUDA-produced output
exists anywhere, so the order `t=b*b; t=fma(a,a,t); t=fma(c,c,t)` used by the
oracle and the HIP kernels is an assumption.  This tool rebuilds the oracle with the two other
orders a compiler could plausibly pick (1: fma(a,a,fma(b,b,c*c)); 2: no contraction) and counts
what changes: FPS indices (a single different pick changes everything after it, so the first
diverging round is reported too), ball-query rows, three_nn picks.

  python tests/fma_order_report.py [--small]      -> table on stdout (profiles/r02_fma_order.txt)
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle  # noqa: E402
from rfdnet_amd import synthetic  # noqa: E402


def run_ops(xyz, plan):
    """plan: list of (n_samples, radius, nsample).  FPS chain + ball query at every level +
    three_nn between consecutive levels."""
    out = {}
    cur = xyz
    levels = [xyz]
    for li, (m, r, ns) in enumerate(plan):
        inds = oracle.furthest_point_sampling(cur, m)
        new = np.ascontiguousarray(cur[:, inds[0]])
        out["fps%d" % (li + 1)] = inds
        out["ball%d" % (li + 1)] = oracle.ball_query(new, cur, r, ns)
        levels.append(new)
        cur = new
    for li in range(len(levels) - 1, 1, -1):
        d2, idx = oracle.three_nn(levels[li - 1], levels[li])
        out["three_nn%d" % (li - 1)] = idx
    return out, levels


def compare(base, other, base_levels):
    rows = []
    for k in base:
        a, b = base[k], other[k]
        if k.startswith("fps"):
            diff = np.where(a[0] != b[0])[0]
            rows.append((k, a.shape[1], len(diff), "first at round %d" % diff[0] if len(diff) else "-"))
        elif k.startswith("ball"):
            bad = (a != b).any(axis=2).sum()
            rows.append((k, a.shape[1], int(bad), "rows"))
        else:
            bad = (a != b).any(axis=2).sum()
            rows.append((k, a.shape[1], int(bad), "points"))
    return rows


def same_input_counts(levels, plan, v):
    """ball query / three_nn on the SAME inputs (variant 0's FPS picks), so that a different FPS
    chain does not mask what the distance test itself changes"""
    rows = []
    for li, (m, r, ns) in enumerate(plan):
        a = oracle.ball_query(levels[li + 1], levels[li], r, ns)
        with oracle.variant(v):
            b = oracle.ball_query(levels[li + 1], levels[li], r, ns)
        rows.append(("ball%d (same centres)" % (li + 1), a.shape[1], int((a != b).any(axis=2).sum()), "rows"))
    for li in range(len(levels) - 1, 1, -1):
        _, a = oracle.three_nn(levels[li - 1], levels[li])
        with oracle.variant(v):
            _, b = oracle.three_nn(levels[li - 1], levels[li])
        rows.append(("three_nn%d (same inputs)" % (li - 1), a.shape[1], int((a != b).any(axis=2).sum()), "points"))
    return rows


def report(scenes, plan):
    lines = []
    for name, xyz in scenes:
        base, levels = run_ops(xyz, plan)
        for v, label in ((1, "fma(a,a,fma(b,b,c*c))"), (2, "no contraction")):
            with oracle.variant(v):
                other, _ = run_ops(xyz, plan)
            lines.append("%s, %d points -- variant %d (%s) vs the assumed order:" % (name, xyz.shape[1], v, label))
            for k, n, bad, what in compare(base, other, levels) + same_input_counts(levels, plan, v):
                lines.append("   %-26s %6d of %6d differ  %s" % (k, bad, n, what))
    return lines


PLAN = [(2048, 0.2, 64), (1024, 0.4, 32), (512, 0.8, 16), (256, 1.2, 16)]     # pointnet2backbone.py:27-61


def main():
    small = "--small" in sys.argv
    scenes = []
    if small:
        pc = synthetic.synthetic_scene(seed=21, n_raw=6000, n_points=4096)
        scenes.append(("F_NET scene", np.ascontiguousarray(pc[None, :, :3])))
    else:
        pc = synthetic.synthetic_scene(seed=10, n_points=80000)
        scenes.append(("config-2 scene (seed 10)", np.ascontiguousarray(pc[None, :, :3])))
        pc = synthetic.synthetic_scene(seed=21, n_raw=6000, n_points=4096)
        scenes.append(("F_NET scene (seed 21)", np.ascontiguousarray(pc[None, :, :3])))
        pc = synthetic.synthetic_scene(seed=10, n_points=40000, n_raw=30000)
        scenes.append(("config-0 scene (with replacement)", np.ascontiguousarray(pc[None, :, :3])))
    for l in report(scenes, PLAN):
        print(l)


if __name__ == "__main__":
    main()
