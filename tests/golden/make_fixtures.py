"""Generate the golden fixtures under tests/golden/ (DEV CONTAINER ONLY).

Imports the reference's own Python from /root/reference (never shipped), runs
it on CPU and stores inputs/outputs as small .npz files.  Weights are NOT
stored: both this script and the tests regenerate them from numpy PCG64 seeds
(rfdnet_amd/synthetic.py), the fixtures keep only the parameter name/shape
lists (which also pin state_dict key parity), inputs and reference outputs.

Recipe = SURVEY.md Appendix A: bare namespace packages so models/__init__.py
(-> loss.py -> chamfer JIT + .cuda()) never executes, `pointnet2_ops._ext`
replaced by the CPU oracle, absent third-party deps stubbed, MISE built from
the reference's mise.pyx with Cython in a scratch directory.

Usage:  python tests/golden/make_fixtures.py [dec] [mise] [grid] [ops] [net] [net80k] [gen] [nms] [cd] [fit] [mc]
"""
import importlib
import os
import subprocess
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from oracle import oracle  # noqa: E402
from rfdnet_amd import synthetic  # noqa: E402


def ns(name, path=None):
    m = types.ModuleType(name)
    m.__path__ = [path] if path else []
    sys.modules[name] = m
    return m


def mount_reference():
    import torch  # noqa: F401
    sys.path.insert(0, REF)
    ns('models', REF + '/models')
    ns('models.iscnet', REF + '/models/iscnet')
    ns('models.iscnet.modules', REF + '/models/iscnet/modules')
    ext = oracle.TorchExt()
    pkg = ns('pointnet2_ops', REF + '/external/pointnet2_ops_lib/pointnet2_ops')
    sys.modules['pointnet2_ops._ext'] = ext
    pkg._ext = ext
    ns('trimesh').Trimesh = lambda *a, **k: a
    ns('mcubes').marching_cubes = lambda vol, thr: (np.zeros((0, 3)), np.zeros((0, 3), int))
    ns('external', REF + '/external')
    ns('external.libsimplify').simplify_mesh = None
    ns('external.libkdtree')
    ns('external.libkdtree.pykdtree')
    ns('external.libkdtree.pykdtree.kdtree').KDTree = None


def build_ref_mise():
    """cythonize the reference's mise.pyx where it lies; output to a scratch dir."""
    d = tempfile.mkdtemp(prefix="refmise_")
    setup = os.path.join(d, "setup.py")
    with open(setup, "w") as f:
        f.write("from setuptools import setup, Extension\n"
                "from Cython.Build import cythonize\nimport numpy\n"
                "setup(ext_modules=cythonize([Extension('mise', ['%s/external/libmise/mise.pyx'],"
                " language='c++', include_dirs=[numpy.get_include()])], build_dir='%s/build',"
                " language_level=3))\n" % (REF, d))
    subprocess.check_call([sys.executable, setup, "build_ext", "--build-lib", d, "--build-temp",
                           d + "/tmp"], cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    sys.path.insert(0, d)
    return importlib.import_module("mise")


def shapes_arrays(shapes):
    names = np.array(list(shapes.keys()))
    shp = np.array([",".join(str(int(x)) for x in s) for s in shapes.values()])
    return names, shp


# ------------------------------------------------------------------ F-DEC ----
def make_dec():
    import torch
    mount_reference()
    occ = importlib.import_module('models.iscnet.modules.occ_decoder')
    torch.manual_seed(0)
    dec = occ.DecoderCBatchNorm(dim=3, z_dim=32, c_dim=512, hidden_size=256)
    shapes = synthetic.load_seeded(dec, seed=1234)
    dec.eval()
    rng = np.random.default_rng(77)
    K, T = 3, 1536
    p = ((rng.random((K, T, 3)) - 0.5) * 1.1).astype(np.float32)
    z = np.zeros((K, 32), dtype=np.float32)
    z[2] = rng.normal(0, 1, 32).astype(np.float32)          # one non-prior z
    c = rng.normal(0, 1, (K, 512)).astype(np.float32)
    with torch.no_grad():
        out = dec(torch.from_numpy(p), torch.from_numpy(z), torch.from_numpy(c)).numpy()
    names, shp = shapes_arrays(shapes)
    np.savez_compressed(os.path.join(HERE, "F_DEC.npz"), seed=1234, names=names, shapes=shp,
                        p=p, z=z, c=c, logits=out.astype(np.float32))
    print("F_DEC: logits", out.shape, float(out.min()), float(out.max()), float(out.std()))


# ----------------------------------------------------------------- F-MISE ----
def sphere_field(pts, res, r=0.35):
    q = pts.astype(np.float64) / res - 0.5
    return r - np.sqrt((q ** 2).sum(-1))


def run_mise(MISE, res0, depth, thr, field):
    m = MISE(res0, depth, thr)
    rounds = []
    p = m.query()
    while p.shape[0] != 0 and len(rounds) < 16:
        v = field(p, m.resolution)
        rounds.append((p.copy(), v.copy()))
        m.update(p, v)
        p = m.query()
    return rounds, m.to_dense()


def make_mise():
    mise = build_ref_mise()
    out = {}
    cases = {
        "sphere_8_2": (8, 2, 0.0, lambda p, r: sphere_field(p, r)),
        "sphere_16_1": (16, 1, 0.0, lambda p, r: sphere_field(p, r)),
        # external/libmise/test.py: MISE(1,2,0.), v = 2*(sum p > 2) - 1
        "libmise_test": (1, 2, 0.0, lambda p, r: 2 * (p.sum(axis=-1) > 2).astype(np.float64) - 1),
        # values exactly at the threshold exercise the non-strict >= / <= (mise.pyx:225-227)
        "plane_eq_4_2": (4, 2, 0.0, lambda p, r: (p[:, 0] - r // 2).astype(np.float64)),
        "empty_4_1": (4, 1, 0.0, lambda p, r: -np.ones(p.shape[0])),
    }
    for name, (res0, depth, thr, field) in cases.items():
        rounds, dense = run_mise(mise.MISE, res0, depth, thr, field)
        out[name + "_cfg"] = np.array([res0, depth, thr], dtype=np.float64)
        out[name + "_nrounds"] = np.array(len(rounds))
        for i, (p, v) in enumerate(rounds):
            out["%s_p%d" % (name, i)] = p.astype(np.int16)
            out["%s_v%d" % (name, i)] = v
        out[name + "_dense"] = dense
        print("F_MISE", name, [r[0].shape[0] for r in rounds], dense.shape)
    # headline shape: only the per-round counts + a checksum (the lists are large)
    rounds, dense = run_mise(mise.MISE, 32, 1, 0.0, lambda p, r: sphere_field(p, r))
    out["sphere_32_1_counts"] = np.array([r[0].shape[0] for r in rounds])
    out["sphere_32_1_dense_sum"] = np.array([dense.sum(), np.abs(dense).sum(), (dense > 0).sum()])
    print("F_MISE sphere_32_1", out["sphere_32_1_counts"])
    np.savez_compressed(os.path.join(HERE, "F_MISE.npz"), **out)


# ----------------------------------------------------------------- F-GRID ----
def make_grid():
    import torch
    mount_reference()
    common = importlib.import_module('external.common')
    out = {}
    for nx in (2, 3, 8, 32, 33):
        g = 1.1 * common.make_3d_grid((-0.5,) * 3, (0.5,) * 3, (nx,) * 3)   # generator.py:92-94
        out["grid_%d" % nx] = g.numpy() if nx <= 8 else g.numpy()[:: max(1, nx * nx * nx // 4096)]
        out["axis_%d" % nx] = (1.1 * torch.linspace(-0.5, 0.5, nx)).numpy()
    np.savez_compressed(os.path.join(HERE, "F_GRID.npz"), **out)
    print("F_GRID ok")


# ------------------------------------------------------------ network level ----
def sub(a, step):
    """strided subset of a flattened array (keeps fixtures small)"""
    return np.ascontiguousarray(a.reshape(-1)[::step])


def make_net():
    """F-NET: the reference's Pointnet2Backbone / VotingModule / ProposalModule /
    SkipPropagation run on CPU on top of the oracle `_ext`."""
    import torch
    mount_reference()
    from rfdnet_amd.iscnet.config import Config
    cfg = Config()
    ns('models.registers')
    reg = importlib.import_module('net_utils.registry')
    sys.modules['models.registers'].MODULES = reg.Registry('module')
    sys.modules['models.registers'].METHODS = reg.Registry('method')
    sys.modules['models.registers'].LOSSES = reg.Registry('loss')
    bbm = importlib.import_module('models.iscnet.modules.pointnet2backbone')
    vm = importlib.import_module('models.iscnet.modules.vote_module')
    pm = importlib.import_module('models.iscnet.modules.proposal_module')
    sp = importlib.import_module('models.iscnet.modules.skip_propagation')
    out = {}
    pc = synthetic.synthetic_scene(seed=21, n_raw=6000, n_points=4096)
    out['pc_seed'] = np.array([21, 6000, 4096])
    x = torch.from_numpy(pc[None])
    with torch.no_grad():
        bb = bbm.Pointnet2Backbone(cfg)
        names, shp = shapes_arrays(synthetic.load_seeded(bb, 101))
        out['bb_names'], out['bb_shapes'] = names, shp
        bb.eval()
        ep = bb(x, {})
        for k in ('sa1_inds', 'sa2_inds', 'fp2_inds'):
            out['bb_' + k] = ep[k].numpy().astype(np.int32)
        for k in ('sa1_xyz', 'sa2_xyz', 'sa3_xyz', 'sa4_xyz'):
            out['bb_' + k] = ep[k].numpy()
        for k in ('sa1_features', 'sa2_features', 'sa3_features', 'sa4_features', 'fp2_features'):
            out['bb_' + k] = ep[k].numpy()                   # FULL tensors (round 2: no strided subsets)
        vote = vm.VotingModule(cfg)
        names, shp = shapes_arrays(synthetic.load_seeded(vote, 102))
        out['vote_names'], out['vote_shapes'] = names, shp
        vote.eval()
        vxyz, vfeat = vote(ep['fp2_xyz'], ep['fp2_features'])
        vfeat = vfeat.div(torch.norm(vfeat, p=2, dim=1).unsqueeze(1))      # demo.py:215-216
        out['vote_xyz'] = vxyz.numpy()
        out['vote_features'] = vfeat.numpy()
        prop = pm.ProposalModule(cfg)
        names, shp = shapes_arrays(synthetic.load_seeded(prop, 103))
        out['prop_names'], out['prop_shapes'] = names, shp
        prop.eval()
        ep['seed_xyz'] = ep['fp2_xyz']
        ep, pf = prop(vxyz, vfeat, ep, True)
        for k in ('aggregated_vote_inds',):
            out['prop_' + k] = ep[k].numpy().astype(np.int32)
        for k in ('aggregated_vote_xyz', 'center', 'objectness_scores', 'heading_scores',
                  'heading_residuals_normalized', 'size_scores', 'size_residuals_normalized',
                  'sem_cls_scores'):
            out['prop_' + k] = ep[k].numpy()
        out['prop_features'] = pf.numpy()
        # skip propagation on 6 of the proposals
        skip = sp.SkipPropagation(cfg)
        names, shp = shapes_arrays(synthetic.load_seeded(skip, 104))
        out['skip_names'], out['skip_shapes'] = names, shp
        skip.eval()
        ids = torch.tensor([[3, 50, 97, 130, 200, 255]])
        centers = torch.gather(ep['center'], 1, ids.unsqueeze(-1).expand(-1, -1, 3))
        feats = torch.gather(pf, 2, ids.unsqueeze(1).expand(-1, 128, -1))
        hc = torch.argmax(ep['heading_scores'], -1)
        res = ep['heading_residuals_normalized'] * (np.pi / 12)
        hr = torch.gather(res, 2, hc.unsqueeze(-1)).squeeze(2)
        ang = cfg.dataset_config.class2angle_cuda(hc, hr)
        ang = torch.gather(ang, 1, ids)
        codes = skip.generate(centers, ang, feats, x)
        out['skip_ids'] = ids.numpy()
        out['skip_angles'] = ang.numpy()
        out['skip_codes'] = codes.numpy()
    np.savez_compressed(os.path.join(HERE, "F_NET.npz"), **out)
    print("F_NET ok", {k: v.shape for k, v in out.items() if not k.endswith(('names', 'shapes'))})


def make_net80k():
    """F-NET80k (round 4): the same three reference modules (pointnet2backbone.py:75-125, vote_module.py,
    proposal_module.py:85-124) on the oracle `_ext` at the size the metric is quoted on -- the headline scene, 80 000
    points (seed 10, 120 000 raw).  Kept: every index tensor, the sampled coordinates, and a 64-point sample of each
    feature tensor (evenly spaced points, all channels) -- < 1 MB."""
    import torch
    mount_reference()
    from rfdnet_amd.iscnet.config import Config
    cfg = Config({'data': {'num_point': 80000}})
    ns('models.registers')
    reg = importlib.import_module('net_utils.registry')
    sys.modules['models.registers'].MODULES = reg.Registry('module')
    sys.modules['models.registers'].METHODS = reg.Registry('method')
    sys.modules['models.registers'].LOSSES = reg.Registry('loss')
    bbm = importlib.import_module('models.iscnet.modules.pointnet2backbone')
    vm = importlib.import_module('models.iscnet.modules.vote_module')
    pm = importlib.import_module('models.iscnet.modules.proposal_module')
    out = {}
    pc = synthetic.synthetic_scene(seed=10, n_raw=120000, n_points=80000)
    out['pc_seed'] = np.array([10, 120000, 80000])
    x = torch.from_numpy(pc[None])

    def sample(t):                       # (1, C, n) -> (C, 64): 64 evenly spaced points, every channel
        n = t.shape[2]
        cols = np.linspace(0, n - 1, 64).astype(np.int64)
        return np.ascontiguousarray(t.numpy()[0][:, cols]), cols.astype(np.int32)

    with torch.no_grad():
        bb = bbm.Pointnet2Backbone(cfg)
        synthetic.load_seeded(bb, 101)
        bb.eval()
        ep = bb(x, {})
        for k in ('sa1_inds', 'sa2_inds', 'fp2_inds'):
            out['bb_' + k] = ep[k].numpy().astype(np.int32)
        for k in ('sa1_xyz', 'sa2_xyz', 'sa3_xyz', 'sa4_xyz'):
            out['bb_' + k] = ep[k].numpy()
        for k in ('sa1_features', 'sa2_features', 'sa3_features', 'sa4_features', 'fp2_features'):
            out['bb_' + k], out['bb_' + k + '_cols'] = sample(ep[k])
        vote = vm.VotingModule(cfg)
        synthetic.load_seeded(vote, 102)
        vote.eval()
        vxyz, vfeat = vote(ep['fp2_xyz'], ep['fp2_features'])
        vfeat = vfeat.div(torch.norm(vfeat, p=2, dim=1).unsqueeze(1))      # demo.py:215-216
        out['vote_xyz'] = vxyz.numpy()
        out['vote_features'], out['vote_features_cols'] = sample(vfeat)
        prop = pm.ProposalModule(cfg)
        synthetic.load_seeded(prop, 103)
        prop.eval()
        ep['seed_xyz'] = ep['fp2_xyz']
        ep, pf = prop(vxyz, vfeat, ep, True)
        out['prop_aggregated_vote_inds'] = ep['aggregated_vote_inds'].numpy().astype(np.int32)
        for k in ('aggregated_vote_xyz', 'center', 'objectness_scores', 'sem_cls_scores'):
            out['prop_' + k] = ep[k].numpy()
        out['prop_features'], out['prop_features_cols'] = sample(pf)
    np.savez_compressed(os.path.join(HERE, "F_NET80k.npz"), **out)
    print("F_NET80k ok", {k: v.shape for k, v in out.items()})


def make_gen():
    """F-GEN: reference ONet + Generator3D (dense 16^3 and MISE 16 -> 32) with the
    MISE built from the reference's mise.pyx; value grids captured at extract_mesh."""
    import torch
    mount_reference()
    mise = build_ref_mise()
    ns('external.libmise').MISE = mise.MISE
    ns('models.registers')
    reg = importlib.import_module('net_utils.registry')
    sys.modules['models.registers'].MODULES = reg.Registry('module')
    sys.modules['models.registers'].METHODS = reg.Registry('method')
    sys.modules['models.registers'].LOSSES = reg.Registry('loss')
    from rfdnet_amd.iscnet.config import Config
    onet_mod = importlib.import_module('models.iscnet.modules.occupancy_net')
    gen_mod = importlib.import_module('models.iscnet.modules.generator')
    out = {}
    rng = np.random.default_rng(5)
    codes = rng.normal(0, 1, (3, 512)).astype(np.float32)
    out['codes'] = codes
    for tag, steps in (('dense16', 0), ('mise16x1', 1), ('mise8x2', 2)):
        res0 = 8 if tag == 'mise8x2' else 16
        cfg = Config({'generation': {'resolution_0': res0, 'upsampling_steps': steps}})
        onet = onet_mod.ONet(cfg)
        shapes = synthetic.load_seeded(onet, 202)
        if tag == 'dense16':
            names, shp = shapes_arrays(shapes)
            out['onet_names'], out['onet_shapes'] = names, shp
        onet.eval()
        grids = []
        gen_mod.Generator3D.extract_mesh = lambda self, occ_hat, z, c=None: grids.append(np.array(occ_hat)) or None
        with torch.no_grad():
            onet.generator.generate_mesh(torch.from_numpy(codes), None)
        g = np.stack(grids)
        out[tag + '_grid'] = g.astype(np.float32)
        assert np.array_equal(g.astype(np.float32).astype(g.dtype), g)
        print("F_GEN", tag, g.shape, float(g.min()), float(g.max()), (g > 0).mean())
    np.savez_compressed(os.path.join(HERE, "F_GEN.npz"), **out)


def make_nms():
    """F-NMS: the reference's own parse_predictions (net_utils/ap_helper.py:131-264,
    run on CPU with Tensor.cuda patched to the identity) + get_proposal_id logic
    (demo.py:50-75) on the proposal head outputs stored in F_NET.npz."""
    import torch
    mount_reference()
    tm = sys.modules['trimesh']
    ex = ns('trimesh.exchange')
    bx = ns('trimesh.exchange.binvox')
    bx.voxelize_mesh = None
    tm.exchange = ex
    ex.binvox = bx
    ap = importlib.import_module('net_utils.ap_helper')
    torch.Tensor.cuda = lambda self, *a, **k: self
    fx = np.load(os.path.join(HERE, "F_NET.npz"))
    seed, n_raw, n_pts = (int(v) for v in fx["pc_seed"])
    pc = synthetic.synthetic_scene(seed=seed, n_raw=n_raw, n_points=n_pts)
    means = np.load(REF + "/datasets/scannet/scannet_means.npz")
    mean_size_arr = means[means.files[0]]
    from rfdnet_amd.iscnet.config import ScannetConfig

    class DC(ScannetConfig):                       # the reference's class2angle / class2size (scannet_config.py:43-73)
        def class2angle(self, pred_cls, residual, to_label_format=True):
            angle = pred_cls * (2 * np.pi / float(self.num_heading_bin)) + residual
            if to_label_format and angle > np.pi:
                angle = angle - 2 * np.pi
            return angle

        def class2size(self, pred_cls, residual):
            return self.mean_size_arr[pred_cls, :] + residual

    dc = DC(mean_size_arr)
    est = {k[5:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("prop_") and k not in
           ("prop_names", "prop_shapes", "prop_features", "prop_aggregated_vote_inds")}
    # make the objectness spread over both sides of 0.5 and classes collide so NMS has work to do
    rng = np.random.default_rng(3)
    est['objectness_scores'] = torch.from_numpy(rng.normal(0, 2.0, (1, 256, 2)).astype(np.float32))
    est['size_residuals_normalized'] = est['size_residuals_normalized'] * 0.2
    out = {'mean_size_arr': mean_size_arr, 'objectness_scores': est['objectness_scores'].numpy(),
           'size_residuals_normalized': est['size_residuals_normalized'].numpy()}
    for tag, cfgd in (("default", {}), ("nocls", {'cls_nms': False}), ("old", {'use_old_type_nms': True}),
                      ("keepempty", {'remove_empty_box': False})):
        config = {'remove_empty_box': True, 'use_3d_nms': True, 'nms_iou': 0.25, 'use_old_type_nms': False,
                  'cls_nms': True, 'per_class_proposal': True, 'conf_thresh': 0.05, 'dataset_config': dc}
        config.update(cfgd)
        eval_dict, parsed = ap.parse_predictions(est, {'point_clouds': torch.from_numpy(pc[None])}, config)
        out[tag + '_pred_mask'] = eval_dict['pred_mask']
        if tag == "default":
            out['corners'] = parsed['pred_corners_3d_upright_camera']
            out['obj_prob'] = parsed['obj_prob']
            prob = torch.softmax(est['objectness_scores'], dim=2)[..., 1].numpy()
            sel = (prob[0] > 0.5) * eval_dict['pred_mask'][0]
            out['proposal_ids'] = np.where(sel.astype(bool))[0]
        print("F_NMS", tag, int(eval_dict['pred_mask'].sum()))
    np.savez_compressed(os.path.join(HERE, "F_NMS.npz"), **out)


def make_cd():
    """F_CD: the reference's OWN CPU Chamfer op (chamfer_distance.cpp, built from the source
    under /root/reference by oracle/build_ref_chamfer.py) on random clouds, an integer lattice
    with many exact ties, and degenerate sizes: inputs + dist/idx both ways + gradients."""
    import torch
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from oracle import build_ref_chamfer
    cd = build_ref_chamfer.load()
    rng = np.random.default_rng(5)
    out = {}
    cases = [("rand", 2, 300, 517), ("lattice", 2, 256, 384), ("single", 1, 7, 1), ("big", 1, 1500, 4099)]
    for name, B, n, m in cases:
        if name == "lattice":
            x1 = rng.integers(0, 4, (B, n, 3)).astype(np.float32)
            x2 = rng.integers(0, 4, (B, m, 3)).astype(np.float32)
        else:
            x1 = rng.standard_normal((B, n, 3)).astype(np.float32)
            x2 = (rng.standard_normal((B, m, 3)) * 1.3 + 0.2).astype(np.float32)
        t1, t2 = torch.from_numpy(x1), torch.from_numpy(x2)
        d1, d2 = torch.zeros(B, n), torch.zeros(B, m)
        i1, i2 = torch.zeros(B, n, dtype=torch.int32), torch.zeros(B, m, dtype=torch.int32)
        cd.forward(t1, t2, d1, d2, i1, i2)
        g1 = rng.standard_normal((B, n)).astype(np.float32)
        g2 = rng.standard_normal((B, m)).astype(np.float32)
        gx1, gx2 = torch.zeros(B, n, 3), torch.zeros(B, m, 3)
        cd.backward(t1, t2, gx1, gx2, torch.from_numpy(g1), torch.from_numpy(g2), i1, i2)
        for k, v in (("x1", x1), ("x2", x2), ("d1", d1.numpy()), ("d2", d2.numpy()), ("i1", i1.numpy()),
                     ("i2", i2.numpy()), ("g1", g1), ("g2", g2), ("gx1", gx1.numpy()), ("gx2", gx2.numpy())):
            out["%s_%s" % (name, k)] = v
        print("F_CD", name, B, n, m)
    out["cases"] = np.array([c[0] for c in cases])
    np.savez_compressed(os.path.join(HERE, "F_CD.npz"), **out)




def make_fit():
    """F_FIT: the reference's own ISCNet.fit_mesh_to_scan (network.py:182-303) run on CPU with
    its Chamfer op = the reference's CPU implementation built from source (cd_ref), on a small
    synthetic scene with three proposals (one filtered by the mask, two fitted).  100 Adam steps
    over padded 10 000 x 50 000 point sets: takes several minutes."""
    import torch
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from oracle import build_ref_chamfer
    cd = build_ref_chamfer.load()

    class _CD(torch.autograd.Function):           # glue only: calls the reference's CPU op
        @staticmethod
        def forward(ctx, a, b):
            a, b = a.contiguous(), b.contiguous()
            d1, d2 = torch.zeros(a.shape[0], a.shape[1]), torch.zeros(b.shape[0], b.shape[1])
            i1 = torch.zeros(a.shape[0], a.shape[1], dtype=torch.int32)
            i2 = torch.zeros(b.shape[0], b.shape[1], dtype=torch.int32)
            cd.forward(a, b, d1, d2, i1, i2)
            ctx.save_for_backward(a, b, i1, i2)
            return d1, d2

        @staticmethod
        def backward(ctx, g1, g2):
            a, b, i1, i2 = ctx.saved_tensors
            ga, gb = torch.zeros_like(a), torch.zeros_like(b)
            cd.backward(a, b, ga, gb, g1.contiguous(), g2.contiguous(), i1, i2)
            return ga, gb

    mount_reference()
    tm = sys.modules['trimesh']
    ex = ns('trimesh.exchange')
    bx = ns('trimesh.exchange.binvox')
    bx.voxelize_mesh = None
    tm.exchange = ex
    ex.binvox = bx
    loss_shim = ns('models.loss')
    loss_shim.chamfer_func = lambda a, b: _CD.apply(a, b)
    net_mod = importlib.import_module('models.iscnet.modules.network')
    from net_utils.box_util import get_3d_box
    from net_utils.libs import flip_axis_to_camera

    rng = np.random.default_rng(21)
    # scene: floor + two box-shaped objects (surface samples, slightly noisy)
    def box_surface(center, size, heading, npts):
        u = rng.uniform(-0.5, 0.5, (npts, 3))
        face = rng.integers(0, 3, npts)
        u[np.arange(npts), face] = np.sign(rng.standard_normal(npts)) * 0.5
        p = u * size
        c, s = np.cos(heading), np.sin(heading)
        R = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])
        return p @ R.T + center
    true_boxes = [((0.8, -0.5, 0.45), (1.2, 0.6, 0.9), 0.3), ((-1.1, 0.9, 0.35), (0.7, 0.7, 0.7), -0.6)]
    floor = np.c_[rng.uniform(-3, 3, (6000, 2)), rng.normal(0, 0.005, 6000)]
    scan = np.concatenate([floor] + [box_surface(np.array(c), np.array(s), h, 4000) for c, s, h in true_boxes])
    scan = scan + rng.normal(0, 0.004, scan.shape)
    scan = np.c_[scan, np.zeros(len(scan))].astype(np.float32)[None]            # (1,N,4)
    # predictions: boxes a bit off the truth (what the fit has to correct); proposal 2 masked out
    pred = [((0.9, -0.42, 0.45), (1.2, 0.6, 0.9), 0.42), ((-1.0, 0.98, 0.35), (0.7, 0.7, 0.7), -0.5),
            ((2.0, 2.0, 0.5), (0.5, 0.5, 1.0), 0.0)]
    K = len(pred)
    corners = np.zeros((1, K, 8, 3))
    for j, (c, s, h) in enumerate(pred):
        corners[0, j] = get_3d_box(np.array(s), -h, flip_axis_to_camera(np.array(c)[None])[0])
    obj_prob = np.array([[0.9, 0.8, 0.95]])
    pred_mask = np.array([[1, 1, 0]])
    parsed = {'pred_corners_3d_upright_camera': corners.copy(), 'pred_sem_cls': np.zeros((1, K), int),
              'obj_prob': obj_prob}
    # meshes: vertices of a unit-ish shape in the generator's canonical frame (any frame: the
    # reference re-centres, permutes axes and normalises extents)
    class M(object):
        pass
    meshes, verts = [], []
    for j in range(K):
        m = M()
        m.vertices = box_surface(np.zeros(3), np.array([0.9, 1.0, 0.8]), 0.0, 1500 + 200 * j) + 0.03
        meshes.append(m)
        verts.append(m.vertices)
    mesh_dict = {'meshes': meshes, 'proposal_ids': np.arange(K).reshape(1, K, 1)}
    dummy = types.SimpleNamespace()
    dummy.chamfer_dist = lambda *a: net_mod.ISCNet.chamfer_dist(dummy, *a)
    out = net_mod.ISCNet.fit_mesh_to_scan(dummy, mesh_dict, parsed, {'pred_mask': pred_mask},
                                           torch.from_numpy(scan), 0.5)
    fitted = out['pred_corners_3d_upright_camera']
    print("F_FIT corner shift (max abs):", np.abs(fitted - corners).reshape(K, -1).max(1))
    sav = {'scan': scan, 'corners_in': corners, 'corners_out': fitted, 'obj_prob': obj_prob,
           'pred_mask': pred_mask, 'n_meshes': np.array(K)}
    for j in range(K):
        sav['verts_%d' % j] = verts[j]
    np.savez_compressed(os.path.join(HERE, "F_FIT.npz"), **sav)


def make_mc():
    """F_MC: the meshes the reference itself ships under demo/outputs/scene0549_00
    (proposal_*_mesh.ply, written by Generator3D.extract_mesh -> PyMCubes 0.1.2 ->
    trimesh export, demo.py:283-287; dense 32^3 grids, canonical coordinates) stored as data:
    float32 vertices and int32 faces exactly as in the files, plus the raw bytes of one PLY
    header.  They are the only marching-cubes output the reference holds."""
    import glob
    from rfdnet_amd import io as rio
    out = {}
    files = sorted(glob.glob(REF + '/demo/outputs/scene0549_00/proposal_*_mesh.ply'))
    assert len(files) == 13
    names = []
    for f in files:
        name = os.path.basename(f)[:-len('_mesh.ply')]
        v, fc = rio.read_mesh_ply(f)
        out[name + '_v'] = np.asarray(v, np.float32)
        out[name + '_f'] = np.asarray(fc, np.int32)
        names.append(name)
    raw = open(files[0], 'rb').read()
    out['ply_header'] = np.frombuffer(raw[:raw.index(b'end_header\n') + len(b'end_header\n')], np.uint8)
    out['ply_header_of'] = np.array(os.path.basename(files[0]))
    out['names'] = np.array(names)
    npz = np.load(REF + '/demo/outputs/scene0549_00/000000_pred_confident_nms_bbox.npz')
    for k in npz.files:                       # the reference-held box dump: keys / dtypes / shapes
        out['bbox_' + k] = npz[k]
    np.savez_compressed(os.path.join(HERE, 'F_MC.npz'), **out)
    print('F_MC.npz', os.path.getsize(os.path.join(HERE, 'F_MC.npz')) // 1024, 'KiB', names)


if __name__ == "__main__":
    what = sys.argv[1:] or ["dec", "mise", "grid"]
    for w in what:
        globals()["make_" + w]()
