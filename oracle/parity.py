"""TEST INFRASTRUCTURE (checker only -- imported by tests/ and by bench.py's after-the-clock parity leg).

The end-to-end parity figure BASELINE.json configs[4] names: the CPU path (oracle decoder -> oracle octree MISE ->
oracle marching cubes, i.e. the reference's Generator3D.generate_from_latent, generator.py:99-117, and extract_mesh,
generator.py:145-168, restated on the host) against the HIP path's value grids and meshes for the same codes.
"""
import numpy as np

from . import oracle


def cpu_value_grid(blob, z, c, resolution0, upsampling_steps, threshold_logit, padding=0.1):
    """One proposal through generator.py:99-117 on the CPU: z (32,), c (C,) -> (value grid float64 (R+1)^3, queries).
    The octree asks for integer lattice points; they become decoder inputs as `box * (p / R - 0.5)` in float32
    (generator.py:106-109: FloatTensor, `/ resolution`, `box_size * (pointsf - 0.5)`); values go back as float64."""
    box = np.float32(1 + padding)
    m = oracle.MISE(resolution0, upsampling_steps, threshold_logit)
    pts = m.query()
    n_q = 0
    zz = np.ascontiguousarray(z, np.float32)[None]
    cc = np.ascontiguousarray(c, np.float32)[None]
    while pts.shape[0]:
        pf = (box * (pts.astype(np.float32) / np.float32(m.resolution) - np.float32(0.5))).astype(np.float32)
        vals = oracle.decoder_cbn(blob, pf[None], zz, cc)[0]
        m.update(pts, vals.astype(np.float64))
        n_q += pts.shape[0]
        pts = m.query()
    return m.to_dense(), n_q


def occupancy_iou(a, b, thr):
    ia, ib = a >= thr, b >= thr
    union = int((ia | ib).sum())
    return (float((ia & ib).sum()) / union if union else 1.0), int((ia != ib).sum())


def vertex_hausdorff(v1, v2):
    """Symmetric Hausdorff distance between two vertex sets (nearest-vertex, scipy k-d tree)."""
    from scipy.spatial import cKDTree
    if len(v1) == 0 and len(v2) == 0:
        return 0.0
    if len(v1) == 0 or len(v2) == 0:
        return float("inf")
    d12 = cKDTree(v2).query(v1)[0].max()
    d21 = cKDTree(v1).query(v2)[0].max()
    return float(max(d12, d21))


def iso_residual(vertices, grid, threshold_logit, padding=0.1):
    """How far off the OTHER path's iso-surface do these mesh vertices lie, in logit units: every marching-cubes vertex
    sits on a grid edge, so the value grid interpolated along that edge at the vertex should equal the threshold.
    vertices: extract_mesh output coordinates (generator.py:163-168: box * ((v_mc - 1.5) / (n - 1) - 0.5) in the
    padded grid's index space); vertices on edges that touch the -1e6 padding shell are skipped.
    -> (max |interpolated value - threshold|, vertices checked)."""
    n = grid.shape[0]
    g = (np.asarray(vertices, np.float64) / (1 + padding) + 0.5) * (n - 1) + 0.5      # unpadded grid index coordinates
    r = np.rint(g)
    axis = np.argmax(np.abs(g - r), axis=1)                                           # the one fractional coordinate
    rows = np.arange(len(g))
    lo = r.astype(np.int64)
    lo[rows, axis] = np.floor(g[rows, axis]).astype(np.int64)
    t = g[rows, axis] - lo[rows, axis]
    hi = lo.copy()
    hi[rows, axis] += 1
    hi[rows, axis] = np.where(t == 0.0, lo[rows, axis], hi[rows, axis])               # a vertex exactly on a lattice point
    ok = (lo >= 0).all(1) & (hi <= n - 1).all(1)
    lo, hi, t = lo[ok], hi[ok], t[ok]
    a = grid[lo[:, 0], lo[:, 1], lo[:, 2]]
    b = grid[hi[:, 0], hi[:, 1], hi[:, 2]]
    val = a + t * (b - a)
    return (float(np.abs(val - threshold_logit).max()) if len(val) else 0.0), int(ok.sum())


def compare(hip_grid, hip_vertices, hip_faces, cpu_grid, threshold_logit, padding=0.1, near=1e-4):
    """-> dict of the parity figures for one proposal.  Vertex distances are in CELLS of the value grid."""
    hip_grid = np.asarray(hip_grid, np.float64)
    iou, flips = occupancy_iou(hip_grid, cpu_grid, threshold_logit)
    n_near = int((np.abs(cpu_grid - threshold_logit) < near).sum())
    cv, cf = oracle.extract_mesh(cpu_grid, threshold_logit, padding)
    cell = (1 + padding) / (hip_grid.shape[0] - 1)
    res, n_res = iso_residual(hip_vertices, np.asarray(cpu_grid, np.float64), threshold_logit, padding)
    return {"iso_residual_logit": res, "iso_vertices_checked": n_res, "iou": iou, "flips": flips, "near_threshold": n_near,
            "max_abs_dlogit": float(np.abs(hip_grid - cpu_grid).max()),
            "points_off_1e-4": int((np.abs(hip_grid - cpu_grid) > 1e-4).sum()),
            "faces_hip": int(len(hip_faces)), "faces_cpu": int(len(cf)),
            "vertices_hip": int(len(hip_vertices)), "vertices_cpu": int(len(cv)),
            "hausdorff_cells": vertex_hausdorff(np.asarray(hip_vertices, np.float64), cv) / cell}
