"""Input / output formats of the demo path.

  read_off            the scan format of demo/inputs/*.off (trimesh.load(...).vertices,
                      demo.py:25)
  load_demo_data      demo.py:24-48: vertices -> [xyz | height] -> random_sampling
                      (utils/pc_util.py:35-47) -> (1, N, 4) float32 tensor
  write_mesh_ply      `proposal_%d_mesh.ply` as trimesh exports it (binary
                      little-endian, float xyz, `list uchar int` faces; header of
                      demo/outputs/scene0549_00/proposal_22_mesh.ply)
  write_points_ply    `%06d_pc.ply` (pc_util.write_ply)
  save_visualization  demo.py:278-327: meshes + `%06d_pred_confident_nms_bbox.npz`
                      with obbs (K,7) and proposal_map (K,1)
"""
import os
import struct

import numpy as np


def read_off(path):
    """-> (vertices (V,3+) float64, faces list).  Handles 'OFF' and 'COFF'
    headers and the 'OFF<counts>' single-line variant."""
    with open(path, "r") as f:
        tokens = f.read().split()
    head = tokens[0]
    pos = 1
    if head not in ("OFF", "COFF", "NOFF"):
        if head.startswith("OFF") and head[3:].isdigit():
            tokens = ["OFF", head[3:]] + tokens[1:]
        else:
            raise ValueError("not an OFF file: %s" % path)
    nv, nf = int(tokens[pos]), int(tokens[pos + 1])
    pos += 3
    rest = tokens[pos:]
    # vertex rows may carry colours: infer the row width from the face section
    faces = []
    width = 3
    for cand in (3, 6, 7):
        if nv * cand <= len(rest):
            p = nv * cand
            ok = True
            for _ in range(min(nf, 8)):
                if p >= len(rest):
                    ok = False
                    break
                k = int(float(rest[p]))
                if k < 1 or k > 64:
                    ok = False
                    break
                p += 1 + k
            if ok:
                width = cand
                break
    v = np.array(rest[:nv * width], dtype=np.float64).reshape(nv, width)
    p = nv * width
    for _ in range(nf):
        k = int(rest[p])
        faces.append([int(x) for x in rest[p + 1:p + 1 + k]])
        p += 1 + k
        # optional per-face colour values are not consumed (unused on this path)
    return v, faces


def random_sampling(pc, num_sample, rng, replace=None):
    """utils/pc_util.py:35-47 (with replacement iff the scan has fewer points)."""
    if replace is None:
        replace = pc.shape[0] < num_sample
    choices = rng.choice(pc.shape[0], num_sample, replace=replace)
    return pc[choices], choices


def load_demo_data(points_or_path, num_point=80000, no_height=False, seed=10):
    """demo.py:24-48 without colours (use_color_* are False in ISCNet_test.yaml)."""
    import torch
    if isinstance(points_or_path, str):
        pc = read_off(points_or_path)[0]
    else:
        pc = np.asarray(points_or_path, dtype=np.float64)
    pc = pc[:, 0:3]
    if not no_height:
        floor_height = np.percentile(pc[:, 2], 0.99)           # 0.99-th percentile, as the reference
        pc = np.concatenate([pc, (pc[:, 2] - floor_height)[:, None]], 1)
    pc, _ = random_sampling(pc, num_point, np.random.default_rng(seed))
    return {'point_clouds': torch.from_numpy(pc.astype(np.float32)[None])}


def write_mesh_ply(path, vertices, faces):
    v = np.asarray(vertices, dtype=np.float32)
    f = np.asarray(faces, dtype=np.int32)
    header = ("ply\nformat binary_little_endian 1.0\ncomment rfdnet_amd\nelement vertex %d\n"
              "property float x\nproperty float y\nproperty float z\nelement face %d\n"
              "property list uchar int vertex_indices\nend_header\n" % (v.shape[0], f.shape[0]))
    rec = np.empty(f.shape[0], dtype=[('n', 'u1'), ('i', '<i4', (3,))])
    rec['n'] = 3
    rec['i'] = f
    with open(path, "wb") as fh:
        fh.write(header.encode("ascii"))
        fh.write(v.astype('<f4').tobytes())
        fh.write(rec.tobytes())


def read_mesh_ply(path):
    with open(path, "rb") as fh:
        data = fh.read()
    end = data.index(b"end_header\n") + len(b"end_header\n")
    head = data[:end].decode("ascii").split("\n")
    nv = int([l for l in head if l.startswith("element vertex")][0].split()[-1])
    nf = int([l for l in head if l.startswith("element face")][0].split()[-1])
    v = np.frombuffer(data, dtype='<f4', count=nv * 3, offset=end).reshape(nv, 3)
    rec = np.frombuffer(data, dtype=[('n', 'u1'), ('i', '<i4', (3,))], count=nf, offset=end + nv * 12)
    return v.copy(), rec['i'].copy()


def write_points_ply(path, points):
    p = np.asarray(points, dtype=np.float32)[:, :3]
    header = ("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\n"
              "property float y\nproperty float z\nend_header\n" % p.shape[0])
    with open(path, "wb") as fh:
        fh.write(header.encode("ascii"))
        fh.write(p.astype('<f4').tobytes())


def save_visualization(output_dir, point_clouds, proposal_ids, meshes, box_params=None, keep_mask=None,
                       batch_id=0):
    """Files of demo.py:278-327.  box_params (K_all,7) = centre, size, heading of
    every proposal; keep_mask (K_all,) selects the dumped ones."""
    os.makedirs(output_dir, exist_ok=True)
    ids = np.asarray(proposal_ids).reshape(-1, 1)
    for mesh, pid in zip(meshes, ids[:, 0]):
        v = mesh.vertices.cpu().numpy() if hasattr(mesh.vertices, "cpu") else mesh.vertices
        f = mesh.faces.cpu().numpy() if hasattr(mesh.faces, "cpu") else mesh.faces
        write_mesh_ply(os.path.join(output_dir, 'proposal_%d_mesh.ply' % int(pid)), v, f)
    write_points_ply(os.path.join(output_dir, '%06d_pc.ply' % batch_id), np.asarray(point_clouds)[batch_id])
    if box_params is not None:
        bp = np.asarray(box_params, dtype=np.float64)
        if keep_mask is not None:
            bp = bp[np.asarray(keep_mask, dtype=bool)]
        np.savez(os.path.join(output_dir, '%06d_pred_confident_nms_bbox.npz' % batch_id),
                 obbs=bp, proposal_map=ids.astype(np.int64))
