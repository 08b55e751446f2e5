"""fit_mesh_to_scan: refine the centre and heading of every detected box so that its generated
mesh lies on the scan points inside the (enlarged) box -- 100 Adam steps on a one-sided Chamfer
loss (models/iscnet/modules/network.py:182-303).  Everything stays on the device: the
nearest-neighbour searches and their gradients are the HIP kernels of csrc/chamfer.hip
(rfdnet_amd.chamfer_distance), the box bookkeeping is batched tensor code.

Reference quirks reproduced on purpose:
  * mesh points and scan points are zero-padded to 10 000 / 50 000 rows and the PADDED mesh
    rows take part in the search (they sit at the box centre after the transform); only the
    scan side is masked in the loss (`mean(dist2 * mask) * 1e3` over all objects at once);
  * the parameters kept are those of the iteration with the lowest loss BEFORE its update;
  * scan points below the 5th height percentile are dropped, a box needs >= 5 points.
Differences: meshes with more than 10 000 vertices are subsampled with a fixed stride (the
reference would fail on the copy), the in-box test is an exact oriented-box test instead of a
Delaunay hull query (identical except for points exactly on a face).
"""
import numpy as np
import torch

from ..chamfer_distance import ChamferDistanceFunction

MAX_OBJ_POINTS = 10000          # network.py:194
MAX_PC_IN_BOX = 50000           # network.py:195
TRANSFORM_SHAPENET = ((0., 0., -1.), (-1., 0., 0.), (0., 1., 0.))      # network.py:191


def flip_axis_to_depth(p):
    """cam (x, y, z) -> depth (x, z, -y)  (net_utils/libs.py:116-120)."""
    return torch.stack([p[..., 0], p[..., 2], -p[..., 1]], -1)


def flip_axis_to_camera(p):
    """depth (x, y, z) -> cam (x, -z, y)  (net_utils/libs.py:98-105)."""
    return torch.stack([p[..., 0], -p[..., 2], p[..., 1]], -1)


def get_3d_box(size, heading, center):
    """size (...,3) [l,w,h], heading (...), center (...,3) -> corners (...,8,3)
    (net_utils/box_util.py:183-198; roty(t) = [[c,0,s],[0,1,0],[-s,0,c]])."""
    dt, dev = size.dtype, size.device
    sx = torch.tensor([1, 1, -1, -1, 1, 1, -1, -1], dtype=dt, device=dev)
    sy = torch.tensor([1, 1, 1, 1, -1, -1, -1, -1], dtype=dt, device=dev)
    sz = torch.tensor([1, -1, -1, 1, 1, -1, -1, 1], dtype=dt, device=dev)
    xc = size[..., 0:1] / 2 * sx
    yc = size[..., 2:3] / 2 * sy
    zc = size[..., 1:2] / 2 * sz
    c, s = torch.cos(heading).unsqueeze(-1), torch.sin(heading).unsqueeze(-1)
    x = c * xc + s * zc
    z = -s * xc + c * zc
    return torch.stack([x + center[..., 0:1], yc + center[..., 1:2], z + center[..., 2:3]], -1)


def box_params_from_corners(corners_cam):
    """(P,8,3) upright-camera corners -> centroid (P,3), sizes (P,3), orientation (P), depth
    frame (network.py:220-229)."""
    d = flip_axis_to_depth(corners_cam)
    centroid = (d.max(dim=1)[0] + d.min(dim=1)[0]) / 2.
    fwd, left, up = d[:, 1] - d[:, 2], d[:, 0] - d[:, 1], d[:, 6] - d[:, 2]
    orientation = torch.atan2(fwd[:, 1], fwd[:, 0])
    sizes = torch.stack([fwd.norm(dim=1), left.norm(dim=1), up.norm(dim=1)], 1)
    return centroid, sizes, orientation


def points_in_box(points, corners):
    """points (N,3), corners (8,3) of a box in get_3d_box order -> bool mask (N)."""
    o = corners[2]
    axes = torch.stack([corners[1] - corners[2], corners[3] - corners[2], corners[6] - corners[2]])   # (3,3)
    t = (points - o) @ axes.t()                      # projections scaled by the edge lengths
    l2 = (axes * axes).sum(1)
    return ((t >= 0) & (t <= l2)).all(dim=1)


def normalise_mesh_points(vertices):
    """network.py:207-211: centre at the bounding-box centre, permute to the ShapeNet frame,
    scale every axis to unit extent."""
    v = vertices.double()
    v = v - (v.max(0)[0] + v.min(0)[0]) / 2.
    v = v @ torch.tensor(TRANSFORM_SHAPENET, dtype=torch.float64, device=v.device).t()
    return v / (v.max(0)[0] - v.min(0)[0])


def chamfer_loss(obj_points, pc_in_box, pc_in_box_masks, centroid_params, orientation_params):
    """network.py:293-303."""
    b_s = obj_points.size(0)
    axis_rectified = torch.zeros(b_s, 3, 3, device=obj_points.device)
    axis_rectified[:, 2, 2] = 1
    axis_rectified[:, 0, 0] = torch.cos(orientation_params)
    axis_rectified[:, 0, 1] = torch.sin(orientation_params)
    axis_rectified[:, 1, 0] = -torch.sin(orientation_params)
    axis_rectified[:, 1, 1] = torch.cos(orientation_params)
    obj_points_after = torch.bmm(obj_points, axis_rectified) + centroid_params.unsqueeze(-2)
    _, dist2 = ChamferDistanceFunction.apply(obj_points_after, pc_in_box)
    return torch.mean(dist2 * pc_in_box_masks) * 1e3


def fit_mesh_to_scan(meshes, proposal_ids, parsed_predictions, eval_dict, input_scan, dump_threshold,
                     lr=0.01, iterations=100):
    """meshes: list of objects with `.vertices` (V,3) for proposal_ids (B,K',1) in order;
    parsed_predictions / eval_dict as returned by predictions.parse_predictions (device tensors or
    numpy); input_scan (B,N,3+) -> parsed_predictions with refined
    'pred_corners_3d_upright_camera' (a copy; the input is not modified)."""
    dev = input_scan.device
    as_t = lambda a, dt: (a if torch.is_tensor(a) else torch.as_tensor(np.asarray(a))).to(dev, dt)
    corners_all = as_t(parsed_predictions['pred_corners_3d_upright_camera'], torch.float64).clone()
    obj_prob = as_t(parsed_predictions['obj_prob'], torch.float64)
    pred_mask = as_t(eval_dict['pred_mask'], torch.int64)
    ids = as_t(proposal_ids, torch.int64)
    bsize, n_prop = obj_prob.shape
    scan = input_scan.double()

    index_list, obj_list, pc_list, mask_list, corner_list = [], [], [], [], []
    sel = ((pred_mask == 1) & (obj_prob > dump_threshold)).cpu().numpy()
    for i in range(bsize):
        id_row = ids[i, :, 0].tolist()
        height = torch.quantile(scan[i, :, 2], 0.05)                         # np.percentile(., 5)
        scene_scan = scan[i, scan[i, :, 2] >= height, :3]
        for j in range(n_prop):
            if not sel[i, j]:
                continue
            verts = meshes[id_row.index(j)].vertices
            verts = verts if torch.is_tensor(verts) else torch.as_tensor(np.asarray(verts))
            verts = verts.to(dev)
            if verts.shape[0] > MAX_OBJ_POINTS:
                stride = -(-verts.shape[0] // MAX_OBJ_POINTS)
                verts = verts[::stride]
            obj_points = normalise_mesh_points(verts)
            centroid, sizes, orientation = box_params_from_corners(corners_all[i, j][None])
            larger = flip_axis_to_depth(get_3d_box(1.2 * sizes, -orientation, flip_axis_to_camera(centroid)))[0]
            inside = scene_scan[points_in_box(scene_scan, larger)]
            if inside.shape[0] < 5:
                continue
            inside = inside[:MAX_PC_IN_BOX]
            om = torch.zeros(MAX_OBJ_POINTS, 3, dtype=torch.float64, device=dev)
            om[:obj_points.shape[0]] = obj_points
            pm = torch.zeros(MAX_PC_IN_BOX, 3, dtype=torch.float64, device=dev)
            pm[:inside.shape[0]] = inside
            mk = torch.zeros(MAX_PC_IN_BOX, dtype=torch.float32, device=dev)
            mk[:inside.shape[0]] = 1
            index_list.append((i, j))
            obj_list.append(om * sizes)                                       # scale to the predicted sizes
            pc_list.append(pm)
            mask_list.append(mk)
            corner_list.append((centroid[0], sizes[0], orientation[0]))
    out = dict(parsed_predictions)
    if not index_list:
        out['pred_corners_3d_upright_camera'] = corners_all
        return out

    obj_points = torch.stack(obj_list).float()
    pc_in_box = torch.stack(pc_list).float()
    pc_masks = torch.stack(mask_list)
    sizes = torch.stack([c[1] for c in corner_list])
    centroid_params = torch.stack([c[0] for c in corner_list]).float().requires_grad_(True)
    orientation_params = torch.stack([c[2] for c in corner_list]).float().requires_grad_(True)
    optimizer = torch.optim.Adam([centroid_params, orientation_params], lr=lr)
    best_c, best_o, best_loss = None, None, 1e6
    with torch.enable_grad():
        for _ in range(iterations):
            optimizer.zero_grad()
            loss = chamfer_loss(obj_points, pc_in_box, pc_masks, centroid_params, orientation_params)
            cur = float(loss.detach())
            if cur < best_loss:
                best_c = centroid_params.detach().clone()
                best_o = orientation_params.detach().clone()
                best_loss = cur
            loss.backward()
            optimizer.step()
    new_corners = get_3d_box(sizes, -best_o.double(), flip_axis_to_camera(best_c.double()))
    for k, (i, j) in enumerate(index_list):
        corners_all[i, j] = new_corners[k]
    out['pred_corners_3d_upright_camera'] = corners_all
    out['fit_loss'] = best_loss
    out['fit_indices'] = index_list
    return out
