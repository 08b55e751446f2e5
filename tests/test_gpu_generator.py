"""GPU (-m gpu): batched MISE state machine, dense grids, generator and marching
cubes."""
import os
import sys

import numpy as np
import pytest
import torch

from rfdnet_amd import synthetic
from rfdnet_amd.iscnet.config import Config

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
TILE = 128


def gpu_mise(hip, fields, res0, depth, thr, active=False):
    """Drive csrc/mise.hip exactly as the generator does, but with analytic
    fields evaluated on the host (so the oracle sees identical values)."""
    lib = hip.lib()
    K = len(fields)
    R1 = (res0 << depth) + 1
    n_per = R1 ** 3
    dev = torch.device("cuda")
    values = torch.zeros(K, n_per, device=dev)
    pstate = torch.empty(K, n_per, dtype=torch.uint8, device=dev)
    vstate = torch.empty(K, lib.rfd_mise_vstate_elems(res0, depth), dtype=torch.uint8, device=dev)
    counts = torch.empty(K, dtype=torch.int32, device=dev)
    dirty = torch.zeros(2, K, lib.rfd_mise_dirty_elems(res0, depth), dtype=torch.uint8, device=dev)
    st = hip.current_stream()
    hip.check(lib.rfd_mise_init(K, res0, depth, pstate.data_ptr(), vstate.data_ptr(), st), "init")
    per_round = []
    for rnd in range(64):
        hip.check(lib.rfd_mise_count(K, res0, depth, pstate.data_ptr(), counts.data_ptr(), st), "count")
        cnt = counts.cpu().numpy().astype(np.int64)
        if cnt.sum() == 0:
            break
        per_round.append(cnt.copy())
        tiles = (cnt + TILE - 1) // TILE
        offs = (np.concatenate([[0], np.cumsum(tiles)[:-1]]) * TILE).astype(np.int32)
        n_tiles = int(tiles.sum())
        tile_prop = torch.from_numpy(np.repeat(np.arange(K, dtype=np.int32), tiles)).cuda()
        pts = torch.zeros(n_tiles * TILE, 3, device=dev)
        lin = torch.full((n_tiles * TILE,), -1, dtype=torch.int32, device=dev)
        cursors = torch.zeros(K, dtype=torch.int32, device=dev)
        hip.check(lib.rfd_mise_collect(K, res0, depth, pstate.data_ptr(), torch.from_numpy(offs).cuda().data_ptr(),
                                       cursors.data_ptr(), 1.1, pts.data_ptr(), lin.data_ptr(), st), "collect")
        l = lin.cpu().numpy()
        p = pts.cpu().numpy()
        tp = np.repeat(tile_prop.cpu().numpy(), TILE)
        coords = np.stack([l // (R1 * R1), (l // R1) % R1, l % R1], 1)
        # the float query points follow generator.py:106-109
        ok = l >= 0
        expect = (np.float32(1.1) * (coords[ok].astype(np.float32) / np.float32(R1 - 1) - np.float32(0.5)))
        np.testing.assert_array_equal(p[ok], expect.astype(np.float32))
        logits = np.zeros(l.shape[0], dtype=np.float32)
        for k in range(K):
            m = ok & (tp == k)
            assert m.sum() == cnt[k]
            logits[m] = fields[k](coords[m], R1 - 1).astype(np.float32)
        hip.check(lib.rfd_mise_scatter(n_tiles, res0, depth, tile_prop.data_ptr(), None, lin.data_ptr(),
                                       torch.from_numpy(logits).cuda().data_ptr(), values.data_ptr(),
                                       pstate.data_ptr(), st), "scatter")
        if active == "dirty" or active == "mixed":
            # the generator's entry point since round 6: dirty-slab bookkeeping; "dirty" = EVERY pass examines only the slabs
            # the round's points touch or the previous pass created voxels in, "mixed" = alternating with full passes
            use = 1 if active == "dirty" else rnd & 1
            hip.check(lib.rfd_mise_subdivide_dirty(K, res0, depth, float(thr), values.data_ptr(), pstate.data_ptr(),
                                                   vstate.data_ptr(), counts.data_ptr(), int(lin.numel()), lin.data_ptr(),
                                                   tile_prop.data_ptr(), dirty[rnd & 1].data_ptr(),
                                                   dirty[(rnd + 1) & 1].data_ptr(), use, st), "subdivide_dirty")
            assert int(dirty[rnd & 1].sum()) == 0                  # the map just used comes back zeroed
        elif active:
            # the generator's entry point until round 5: a proposal whose query was empty this round is finished (the reference ends
            # that object's loop, generator.py:104) and is skipped; `counts` = what this round evaluated
            hip.check(lib.rfd_mise_subdivide_active(K, res0, depth, float(thr), values.data_ptr(), pstate.data_ptr(),
                                                    vstate.data_ptr(), counts.data_ptr(), st), "subdivide_active")
        else:
            hip.check(lib.rfd_mise_subdivide(K, res0, depth, float(thr), values.data_ptr(), pstate.data_ptr(),
                                             vstate.data_ptr(), st), "subdivide")
    hip.check(lib.rfd_mise_to_dense(K, res0, depth, values.data_ptr(), pstate.data_ptr(), st), "dense")
    return values.view(K, R1, R1, R1).cpu().numpy(), per_round


def oracle_mise(oracle, field, res0, depth, thr):
    m = oracle.MISE(res0, depth, thr)
    counts = []
    p = m.query()
    while p.shape[0]:
        counts.append(p.shape[0])
        m.update(p, field(p, m.resolution).astype(np.float32).astype(np.float64))
        p = m.query()
    return m.to_dense(), counts


def sphere(r, c=(0.5, 0.5, 0.5)):
    return lambda p, R: r - np.sqrt(((p.astype(np.float64) / R - np.array(c)) ** 2).sum(-1))


def two_blobs(p, R):
    q = p.astype(np.float64) / R
    return np.maximum(0.18 - np.linalg.norm(q - [0.3, 0.3, 0.35], axis=-1),
                      0.12 - np.linalg.norm(q - [0.72, 0.66, 0.6], axis=-1))


def thin_slab(p, R):      # sheet thinner than a coarse voxel: only found through neighbours' splits
    q = p.astype(np.float64) / R
    return 0.02 - np.abs(q[:, 0] - 0.53) - 0.3 * np.abs(q[:, 1] - 0.5)


def plane_on_threshold(p, R):   # exact zeros: exercises the non-strict >= / <= (mise.pyx:225-227)
    return (p[:, 0] - R // 2).astype(np.float64)


@pytest.mark.parametrize("active", [False, True, "dirty", "mixed"])
@pytest.mark.parametrize("res0,depth", [(4, 1), (8, 2), (16, 1), (4, 3), (32, 1), (32, 2)])
def test_batched_mise_equals_octree_oracle(hip, oracle, res0, depth, active):
    """(32, 1) = the headline 64^3; (32, 2) = the 128^3 sweep configuration (configs[4]).  active: through
    rfd_mise_subdivide_active, which skips the proposals that evaluated nothing this round (the six fields finish after
    one to five rounds, so every round of the deeper cases has finished and unfinished proposals side by side).  "dirty" /
    "mixed": through rfd_mise_subdivide_dirty with every pass (every other pass) restricted to the dirty slabs."""
    fields = [sphere(0.35), two_blobs, thin_slab, plane_on_threshold,
              lambda p, R: -np.ones(p.shape[0]), sphere(0.2, (0.4, 0.55, 0.6))]
    dense, rounds = gpu_mise(hip, fields, res0, depth, 0.0, active=active)
    if active:
        assert len({sum(1 for r in rounds if r[k] > 0) for k in range(len(fields))}) > 1 or depth == 1
    for k, f in enumerate(fields):
        ref, counts = oracle_mise(oracle, f, res0, depth, 0.0)
        np.testing.assert_array_equal(dense[k].astype(np.float64), ref)
        assert [int(r[k]) for r in rounds if r[k] > 0] == counts       # same per-round query counts


def test_batched_mise_nonzero_threshold(hip, oracle):
    thr = float(np.log(0.2) - np.log(0.8))
    fields = [lambda p, R: 3 * sphere(0.3)(p, R) - 1.0, two_blobs]
    dense, _ = gpu_mise(hip, fields, 8, 2, thr)
    for k, f in enumerate(fields):
        ref, _ = oracle_mise(oracle, f, 8, 2, thr)
        np.testing.assert_array_equal(dense[k].astype(np.float64), ref)


def test_dense_grid_points_match_oracle(hip, oracle):
    lib = hip.lib()
    for n in (2, 3, 8, 32, 33):
        npad = (n ** 3 + TILE - 1) // TILE * TILE
        pts = torch.empty(npad, 3, device="cuda")
        hip.check(lib.rfd_make_grid_points(n, -0.5, 0.5, 1.1, pts.data_ptr(), npad, hip.current_stream()), "grid")
        np.testing.assert_array_equal(pts.cpu().numpy()[: n ** 3], oracle.make_3d_grid(-0.5, 0.5, n, 1.1))
        assert float(pts[n ** 3:].abs().sum()) == 0.0


@pytest.fixture(scope="module")
def onet_and_fixture(hip, golden_dir):
    fx = np.load(os.path.join(golden_dir, "F_GEN.npz"))
    return fx


def make_onet(res0, steps):
    from rfdnet_amd.iscnet.occupancy_net import ONet
    onet = ONet(Config({'generation': {'resolution_0': res0, 'upsampling_steps': steps}}))
    # the reference ONet owns encoder_latent.* too: seed in the reference's key order
    return onet


def load_onet_seeded(onet, fx, seed=202):
    from collections import OrderedDict
    shapes = OrderedDict((str(n), tuple(int(x) for x in str(s).split(",")) if str(s) else ())
                         for n, s in zip(fx["onet_names"], fx["onet_shapes"]))
    sd = synthetic.seeded_state_dict(shapes, seed)
    own = onet.state_dict()
    onet.load_state_dict({k: torch.from_numpy(sd[k]) for k in own})
    return onet.cuda().eval()


def test_generator_dense_grid_matches_reference(hip, onet_and_fixture):
    fx = onet_and_fixture
    onet = load_onet_seeded(make_onet(16, 0), fx)
    grids = onet.generator.generate_grids(torch.from_numpy(fx["codes"]).cuda(), None)
    hip.device_status()
    assert grids.shape == (3, 16, 16, 16)
    assert np.abs(grids.cpu().numpy() - fx["dense16_grid"]).max() < 1e-4


@pytest.mark.parametrize("tag,res0,steps", [("mise16x1", 16, 1), ("mise8x2", 8, 2)])
def test_generator_mise_grid_matches_reference(hip, onet_and_fixture, tag, res0, steps):
    """Reference Generator3D + compiled mise.pyx vs the batched device pipeline.
    MISE is data dependent: a logit within 1e-6 of the threshold may flip a
    subdivision, so a vanishing fraction of fine points may be filled instead of
    evaluated -- everything else must agree to the logit tolerance."""
    fx = onet_and_fixture
    onet = load_onet_seeded(make_onet(res0, steps), fx)
    grids = onet.generator.generate_grids(torch.from_numpy(fx["codes"]).cuda(), None).cpu().numpy()
    hip.device_status()
    ref = fx[tag + "_grid"]
    assert grids.shape == ref.shape
    bad = np.abs(grids - ref) > 1e-4
    flips = (grids > 0) != (ref > 0)
    print("%s: %d of %d grid points differ by > 1e-4 (max |d| %.3g), %d sign flips"
          % (tag, int(bad.sum()), bad.size, float(np.abs(grids - ref).max()), int(flips.sum())))
    # observed on MI355X (rounds 2 and 3): 0 of 107 811 points off, 0 flips; one flipped subdivision would move at most
    # one coarse voxel's worth of fine points (<= 27 at one upsampling step) -- allow exactly that much
    assert int(bad.sum()) <= 27, int(bad.sum())
    assert int(flips.sum()) <= 1, int(flips.sum())


def test_scatter_fused_into_the_decoder_gives_the_same_grids(hip, onet_and_fixture):
    """rfd_occ_decode_scatter_w8 (MISE.update's value / known part in the decoder's epilogue) against
    rfd_occ_decode_w8 + rfd_mise_scatter: identical value grids, bit for bit, incl. the shared round-0 list"""
    fx = onet_and_fixture
    codes = torch.from_numpy(fx["codes"]).cuda()
    out = []
    for fuse in (True, False):
        onet = load_onet_seeded(make_onet(16, 1), fx)
        assert onet.decoder.kernel == "w8"
        onet.decoder.fuse_scatter = fuse
        assert onet.decoder.can_scatter() == fuse
        out.append(onet.generator.generate_grids(codes, None))
        hip.device_status()
    assert torch.equal(out[0], out[1])


# ------------------------------------------------------------ marching cubes ----
from mc_ref import canon, marching_cubes_soup as np_marching_cubes_soup  # noqa: E402


def field_grid(n, fn):
    g = np.stack(np.meshgrid(*[np.arange(n)] * 3, indexing="ij"), -1).reshape(-1, 3)
    return fn(g, n - 1).reshape(n, n, n).astype(np.float32)


def test_marching_cubes_matches_numpy_restatement_and_is_watertight(hip, oracle):
    from rfdnet_amd.iscnet.mcubes import marching_cubes_batch
    n = 12
    rng = np.random.default_rng(0)
    grids = np.stack([field_grid(n, sphere(0.33)), field_grid(n, two_blobs),
                      rng.normal(size=(n, n, n)).astype(np.float32),         # every case incl. ambiguous
                      np.full((n, n, n), -1, dtype=np.float32),              # empty
                      np.full((n, n, n), 1, dtype=np.float32)])              # solid: closed by the padding
    onlevel = rng.normal(size=(n, n, n)).astype(np.float32)
    onlevel[rng.random((n, n, n)) < 0.2] = 0.0                               # values ON the iso level
    grids = np.concatenate([grids, onlevel[None]])
    out = marching_cubes_batch(torch.from_numpy(grids).cuda(), 0.0)
    for k in range(grids.shape[0]):
        v, f = out[k][0].cpu().numpy(), out[k][1].cpu().numpy()
        # the oracle = the library's algorithm incl. its vertex / triangle ORDER (pinned by the
        # reference's demo meshes, tests/test_mcubes_golden.py): identical index arrays
        ov, of = oracle.marching_cubes(np.pad(grids[k].astype(np.float64), 1, constant_values=-1e6), 0.0)
        assert np.array_equal(f, of), k
        assert v.shape == ov.shape and (v.size == 0 or np.abs(v - ov).max() < 1e-12), k
        ref = np_marching_cubes_soup(grids[k], 0.0)
        assert f.shape[0] == ref.shape[0]
        if f.shape[0] == 0:
            continue
        if k == grids.shape[0] - 1:
            continue        # degenerate (zero-area) triangles at on-level points: soup checks do not apply
        assert canon(v[f]) == canon(ref)
        # shared vertices: every vertex is used, no duplicates
        assert len(np.unique(f)) == v.shape[0] and len(np.unique(np.round(v, 9), axis=0)) == v.shape[0]
        # closed, consistently oriented 2-manifold: each directed edge once, its reverse once
        e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
        de = set(map(tuple, e))
        assert len(de) == e.shape[0]
        assert all((b, a) in de for a, b in de)
    # outward normals: positive signed volume for the solid sphere, ~ (4/3) pi r^3
    v, f = out[0][0].cpu().numpy(), out[0][1].cpu().numpy()
    vol = np.einsum('ij,ij->i', v[f[:, 0]], np.cross(v[f[:, 1]], v[f[:, 2]])).sum() / 6
    r = 0.33 * (n - 1)
    assert 0.85 * 4 / 3 * np.pi * r ** 3 < vol < 1.05 * 4 / 3 * np.pi * r ** 3


def test_marching_cubes_reproduces_the_reference_demo_meshes(hip):
    """The 13 meshes the reference ships (demo/outputs/scene0549_00, F_MC.npz): value grids
    rebuilt from them go through the HIP marching cubes + Generator3D.extract_meshes and must
    give back the same FACE ARRAYS (order, indexing, rotation) and the same vertices
    (float32-PLY precision; the reference's -1.5 transform)."""
    import mc_golden as MG
    from rfdnet_amd.iscnet.generator import Generator3D
    z, names = MG.load()
    grids = []
    for nm in names:
        u, base, axis, t, inside = MG.analyse(z[nm + "_v"], z[nm + "_f"])
        grids.append(MG.rebuild_grid(base, axis, t, inside))
    gen = Generator3D(None, threshold=0.5, resolution0=32, upsampling_steps=0, padding=0.1)
    meshes = gen.extract_meshes(torch.from_numpy(np.stack(grids)).cuda())
    worst = 0.0
    for nm, m in zip(names, meshes):
        v, f = m.vertices.cpu().numpy(), m.faces.cpu().numpy()
        assert np.array_equal(f, z[nm + "_f"]), nm
        err = np.abs(v - z[nm + "_v"].astype(np.float64)).max() * 31 / 1.1
        worst = max(worst, err)
        assert err < 2e-3, (nm, err)
    print("HIP marching cubes vs reference demo meshes: faces identical, max vertex deviation %.2e cells" % worst)


def test_generate_mesh_end_to_end_scaling(hip, onet_and_fixture):
    """vertices land where generator.py:163-168 puts them: on edges of the lattice shifted by
    -1.5 padded cells, inside [-0.55 (1 + 1/(n-1)), 0.55 (1 - 1/(n-1))]"""
    fx = onet_and_fixture
    onet = load_onet_seeded(make_onet(16, 1), fx)
    meshes = onet.generator.generate_mesh(torch.from_numpy(fx["codes"]).cuda(), None)
    assert len(meshes) == 3
    for m in meshes:
        v = m.vertices.cpu().numpy()
        assert v.shape[0] > 0 and m.faces.shape[1] == 3
        assert v.min() >= -0.55 * (1 + 1 / 32) - 1e-6 and v.max() <= 0.55 * (1 - 1 / 32) + 1e-6   # -1e6 padding: 3e-6 cells
        u = (v / 1.1 + 0.5) * 32 + 1.5                       # n = 33 grid points per axis
        onlat = np.abs(u - np.round(u)) < 1e-6
        assert (onlat.sum(1) >= 2).all()
        assert int(m.faces.max()) < v.shape[0]


@pytest.mark.parametrize("scale", [1.0, 3.5, 5.0, 5.6, 6.5])
def test_mise_path_in_the_logit_bands_of_a_trained_checkpoint(hip, oracle, onet_and_fixture, scale):
    """The logit bands of tests/test_gpu_decoder.py once through the whole MISE loop (generator.py:99-117): conditioning
    codes scaled so that |logit| reaches ~1 / 3 / 10 / 30, fc_out.bias shifted so that the threshold cuts through the
    middle of each proposal-0 field (otherwise the scaled fields are one-signed and MISE never refines); HIP value grids
    against the CPU path (oracle decoder -> octree oracle, oracle/parity.py).  A decision may flip only where the field
    is within fp32 noise of the threshold."""
    from collections import OrderedDict
    from oracle import parity
    fx = onet_and_fixture
    res0, steps = 16, 1
    onet = load_onet_seeded(make_onet(res0, steps), fx)
    gen = onet.generator
    thr = gen.logit_threshold()
    codes_h = (fx["codes"] * np.float32(scale)).astype(np.float32)
    z = onet.get_z_from_prior((codes_h.shape[0],), sample=gen.sample, device="cpu").numpy()
    state = lambda: OrderedDict((k, v.detach().cpu().numpy()) for k, v in onet.decoder.state_dict().items())
    lattice = oracle.make_3d_grid(-0.5, 0.5, res0 + 1, 1.0 + gen.padding)
    level0 = oracle.decoder_cbn(oracle.decoder_param_blob(state()), lattice[None], z[:1], codes_h[:1])
    with torch.no_grad():
        onet.decoder.fc_out.bias.sub_(float(np.median(level0)) - thr)
    blob = oracle.decoder_param_blob(state())
    codes = torch.from_numpy(codes_h).cuda()
    grids = gen.generate_grids(codes, None).cpu().numpy()
    hip.device_status()
    flips = near = queries = off = 0
    worst = top = 0.0
    cpu = []
    for k in range(codes.shape[0]):
        cpu_grid, n_q = parity.cpu_value_grid(blob, z[k], codes_h[k], res0, steps, thr, gen.padding)
        cpu.append(cpu_grid)
        queries += n_q
        top = max(top, float(np.abs(cpu_grid).max()))
    # fp32 noise of the oracle itself grows with the activations (tests/test_gpu_decoder.py bands): 1e-4 up to |logit| 10
    tol = 1e-4 if top <= 10 else 1e-5 * top
    for k, cpu_grid in enumerate(cpu):
        flips += int(((grids[k] >= thr) != (cpu_grid >= thr)).sum())
        near += int((np.abs(cpu_grid - thr) < 1e-4).sum())
        d = np.abs(grids[k] - cpu_grid)
        off += int((d > tol).sum())
        worst = max(worst, float(d[d <= tol].max()))
    print("codes x%.1f: |logit| up to %.1f, %d query points (HIP %d), max |HIP grid - CPU grid| %.2e over the points "
          "within %.0e, %d points beyond it, %d flipped decisions (%d points within 1e-4 of the threshold)"
          % (scale, top, queries, gen.stats['n_queries'], worst, tol, off, flips, near))
    # MISE is data dependent: a value within fp32 noise of the threshold may flip ONE subdivision, and then up to 27 fine
    # points of that coarse voxel are evaluated on one side and filled from the coarser level on the other (observed on
    # MI355X at codes x6.5: 1 flip with 2 points within 1e-4 of the threshold; none in the other bands)
    assert flips <= max(1, near)
    # (measured: 87 points for the one flip at codes x6.5 -- the 19 new points of the voxel plus what to_dense's
    # forward fill along x, y, z copies from them)
    assert off <= 125 * min(flips, near), (off, flips, near)
    assert queries == gen.stats['n_queries'] or flips
