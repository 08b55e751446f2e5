"""Batched marching cubes on the device (csrc/mcubes.hip): the counterpart of
`mcubes.marching_cubes(np.pad(occ_hat, 1, 'constant', constant_values=-1e6),
threshold)` in Generator3D.extract_mesh (generator.py:157-161), for all K
proposals at once."""
import torch

from .. import _lib


def _call(name, dev, *args):
    with torch.cuda.device(dev):
        rc = getattr(_lib.lib(), name)(*args, _lib.current_stream())
    _lib.check(rc, name)


@torch.no_grad()
def marching_cubes_batch(grids, threshold, pad_value=-1e6, return_flat=False, affine=None):
    """grids (K,n,n,n) f32 device tensor -> list of K (vertices (nv,3) f64,
    faces (nt,3) i32) device tensors.  Vertex coordinates are in the index
    space of the PADDED grid (original grid point i at i + 1).
    return_flat=True -> (all vertices, all faces, vertex bounds, face bounds):
    one buffer each for the K meshes, split points as Python lists.
    affine=(a, c): the stored vertices are a * v + c (one fma per coordinate inside the emitting kernel)."""
    assert grids.is_cuda and grids.dtype == torch.float32 and grids.dim() == 4
    grids = grids.contiguous()
    K, n = grids.shape[0], grids.shape[1]
    assert grids.shape[2] == n and grids.shape[3] == n
    dev = grids.device
    D = n + 2
    per = D * D * D
    nblk = _lib.lib().rfd_mc_blocks(n)
    code = torch.empty(K * per, dtype=torch.uint8, device=dev)
    sums = torch.empty(2, K * nblk, dtype=torch.int32, device=dev)          # vertices / triangles
    _call("rfd_mc_classify", dev, K, n, float(pad_value), float(threshold), grids.data_ptr(),
          code.data_ptr(), sums[0].data_ptr(), sums[1].data_ptr())
    # one 1-D scan over both rows (the 2-row innermost-dim scan kernel is ~70x slower)
    flat = torch.cumsum(sums.view(-1), 0, dtype=torch.int32)
    inc = flat.view(2, -1) - torch.stack([flat.new_zeros(()), flat[K * nblk - 1]]).unsqueeze(1)
    base = inc - sums
    # per-proposal boundaries + totals: one small D2H copy
    bounds = inc[:, nblk - 1::nblk].cpu()
    vend = [0] + bounds[0].tolist()
    tend = [0] + bounds[1].tolist()
    nv, nt = vend[-1], tend[-1]
    verts = torch.empty(max(nv, 1), 3, dtype=torch.float64, device=dev)
    tris = torch.empty(max(nt, 1), 3, dtype=torch.int32, device=dev)
    if nv:
        vbase = torch.empty(K * per, dtype=torch.int32, device=dev)          # scratch
        va, vc = (1.0, 0.0) if affine is None else (float(affine[0]), float(affine[1]))
        _call("rfd_mc_emit_affine", dev, K, n, float(pad_value), float(threshold), grids.data_ptr(),
              code.data_ptr(), base[0].data_ptr(), base[1].data_ptr(), vbase.data_ptr(),
              verts.data_ptr(), tris.data_ptr(), va, vc)
    if return_flat:
        return verts[:nv], tris[:nt], vend, tend
    return [(verts[vend[k]:vend[k + 1]], tris[tend[k]:tend[k + 1]]) for k in range(K)]
