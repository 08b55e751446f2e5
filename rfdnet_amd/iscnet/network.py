"""ISCNet: sub-networks assembled by name from the config's `model:` block and
the generation pipeline of the reference's demo / test drivers.

  ISCNet.__init__   models/iscnet/modules/network.py:17-54 (phase -> attribute
                    names backbone, voting, detection, skip_propagation,
                    completion; classes looked up in MODULES)
  ISCNet.generate   demo.py:200-276 `generate` / network.py:56-180 (minus the
                    GT-dependent steps, which need datasets absent here)
  load_weight       models/network.py:81-89 (strip `module.`, module-by-module,
                    missing keys tolerated)

Proposal selection.  'nms' is the reference's behaviour: proposals with
objectness > dump_threshold that survive empty-box removal and class-aware 3-D
NMS (parse_predictions, ap_helper.py:131-264 -- CPU numpy + scipy there, device
kernels here, see predictions.py).  'all' keeps every one of the num_target
proposals (BASELINE configs 1-4 are quoted on "256 proposals"); 'objectness'
applies the probability threshold only.
"""
import numpy as np
import torch
import torch.nn as nn

from . import (occupancy_net, pointnet2backbone, proposal_module,  # noqa: F401  (register)
               skip_propagation, vote_module)
from .registers import METHODS, MODULES


class _StageFlag(Exception):
    """status bits raised by the stages in front of the completion (internal to ISCNet.reconstruct / complete)"""

    def __init__(self, status):
        super().__init__("status %d" % status)
        self.status = status


@METHODS.register_module
class ISCNet(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        phase = cfg.config[cfg.config['mode']]['phase']
        phase_names = []
        if phase in ['detection']:
            phase_names += ['backbone', 'voting', 'detection']
        if phase in ['completion']:
            phase_names += ['backbone', 'voting', 'detection', 'completion']
            if cfg.config['data']['skip_propagate']:
                phase_names += ['skip_propagation']
        if not cfg.config.get('model') or not phase_names:
            raise ModuleNotFoundError('No submodule found. Please check the phase name and model definition.')
        for phase_name, net_spec in cfg.config['model'].items():
            if phase_name not in phase_names:
                continue
            cls = MODULES.get(net_spec['method'])
            if cls is None:
                raise ModuleNotFoundError('unknown module %r' % net_spec['method'])
            self.add_module(phase_name, cls(cfg, None))

    def load_weight(self, pretrained_model):
        """state_dict with the reference's key names (optionally `module.`-prefixed)."""
        own = self.state_dict()
        stripped = {'.'.join(k.split('.')[1:]) if k.startswith('module.') else k: v
                    for k, v in pretrained_model.items()}
        own.update({k: v for k, v in stripped.items() if k in own})
        self.load_state_dict(own)

    def worker_view(self):
        """A second handle on THIS network for another host thread / stream: every parameter, buffer, sub-module and
        packed-weight cache is shared (read-only at inference); only the mesh generator -- the one object that keeps
        per-call state (query statistics, the mesh buffers of the last call, the round hook) -- is the view's own.
        Several scenes in flight per GPU then cost one set of weights, not one per scene."""
        import copy
        view = copy.copy(self)
        view._modules = dict(self._modules)              # a dict of its own: re-pointing `completion` stays local
        comp = copy.copy(self.completion)
        comp.__dict__ = dict(self.completion.__dict__)
        gen = copy.copy(self.completion.generator)
        gen.__dict__ = dict(self.completion.generator.__dict__)
        gen.model, gen.stats, gen.round_hook = comp, {}, None
        gen.__dict__.pop('last_buffers', None)
        # the round-0 lattice cache is cleared wholesale when full; a view sharing the dict could free tensors another
        # stream's decode still reads (torch's allocator tracks the allocating stream only): every view its own
        gen.__dict__.pop('_round0_cache', None)
        comp.generator = gen
        view._modules['completion'] = comp
        return view

    # ---------------------------------------------------------------- stages ---
    def detect(self, point_clouds):
        """backbone -> voting (+L2 norm) -> proposal  (demo.py:206-221)."""
        end_points = self.backbone(point_clouds, {})
        xyz, features = end_points['fp2_xyz'], end_points['fp2_features']
        end_points['seed_inds'] = end_points['fp2_inds']
        end_points['seed_xyz'] = xyz
        end_points['seed_features'] = features
        xyz, features = self.voting(xyz, features)
        features = features.div(torch.norm(features, p=2, dim=1).unsqueeze(1))
        end_points['vote_xyz'] = xyz
        end_points['vote_features'] = features
        end_points, proposal_features = self.detection(xyz, features, end_points, True)
        return end_points, proposal_features

    def select_proposals(self, end_points, selection='all', point_clouds=None):
        """-> (B, K, 1) int64 proposal ids (the layout of BATCH_PROPOSAL_IDs, demo.py:50-75)."""
        B, P = end_points['center'].shape[0], end_points['center'].shape[1]
        dev = end_points['center'].device
        if selection == 'all':
            return torch.arange(P, device=dev).view(1, P, 1).expand(B, P, 1).contiguous()
        if selection == 'nms':      # the reference's behaviour (demo.py:223-234)
            from . import predictions
            eval_dict, parsed = predictions.parse_predictions(
                end_points, point_clouds, self.cfg.dataset_config, getattr(self.cfg, 'eval_overrides', None))
            end_points['parsed_predictions'] = parsed
            end_points['pred_mask'] = eval_dict['pred_mask']
            return predictions.get_proposal_id(end_points, eval_dict['pred_mask'],
                                               self.cfg.config['generation']['dump_threshold'])
        if selection == 'objectness':
            assert B == 1
            thr = self.cfg.config['generation']['dump_threshold']
            prob = torch.softmax(end_points['objectness_scores'], dim=2)[..., 1]
            ids = torch.nonzero(prob[0] > thr).view(1, -1, 1)
            return ids
        raise ValueError(selection)

    def object_codes(self, end_points, proposal_features, ids, point_clouds):
        """gather per-proposal features / centres / headings and run skip
        propagation (demo.py:236-258) -> (B*K, c_dim)."""
        dev = end_points['center'].device
        idx = ids[..., 0]
        feats = torch.gather(proposal_features, 2, idx.unsqueeze(1).expand(-1, 128, -1))
        if not self.cfg.config['data']['skip_propagate']:
            obj = feats
        else:
            centers = torch.gather(end_points['center'], 1, idx.unsqueeze(-1).expand(-1, -1, 3))
            dc = self.cfg.dataset_config
            heading_class = torch.argmax(end_points['heading_scores'], -1)
            residuals = end_points['heading_residuals_normalized'] * (np.pi / dc.num_heading_bin)
            heading_residual = torch.gather(residuals, 2, heading_class.unsqueeze(-1)).squeeze(2)
            angles = dc.class2angle_cuda(heading_class, heading_residual)
            angles = torch.gather(angles, 1, idx)
            obj = self.skip_propagation.generate(centers.contiguous(), angles, feats, point_clouds)
        B, C, K = obj.size()
        return obj.transpose(1, 2).contiguous().view(B * K, C)

    def cls_codes(self, end_points, ids):
        sem = end_points['sem_cls_scores']
        g = torch.gather(sem, 1, ids[..., 0].unsqueeze(-1).expand(-1, -1, sem.size(2)))
        one_hot = (g >= torch.max(g, dim=2, keepdim=True)[0]).float()
        return one_hot.view(-1, sem.size(2))

    @torch.no_grad()
    def fit_mesh_to_scan(self, pred_mesh_dict, parsed_predictions, eval_dict, input_scan, dump_threshold):
        """Box refinement of the evaluation path (network.py:182-303); see iscnet/fit.py.
        pred_mesh_dict = {'meshes': [...], 'proposal_ids': (B,K',1)} as in the reference."""
        from . import fit
        return fit.fit_mesh_to_scan(pred_mesh_dict['meshes'], pred_mesh_dict['proposal_ids'], parsed_predictions,
                                    eval_dict, input_scan, dump_threshold)

    def generate(self, data, selection='all', return_grids=False):
        """data['point_clouds'] (B,N,3+f) -> (end_points, proposal ids, meshes)."""
        pc = data['point_clouds']
        end_points, proposal_features = self.detect(pc)
        ids = self.select_proposals(end_points, selection, pc)
        if ids.shape[1] == 0:                  # nothing survived the selection
            # detect() may have raised a flag (FPS abort): it is this scene's, read it before returning (reconstruct()
            # does the same on the other path)
            from .. import _lib
            _lib.raise_status(_lib.stream_status_bits())
            return end_points, ids, []
        out = self.reconstruct(end_points, proposal_features, ids, pc, return_grids=return_grids)
        return end_points, ids, out

    def reconstruct(self, end_points, proposal_features, ids, pc, return_grids=False, hook=None):
        """skip propagation -> object codes -> occupancy completion for the selected proposals, with the status
        reads that must follow (see complete()).  A split-precision GEMM activation beyond the f16 range at the
        default scale (status bit 4, skip-propagation encoder) does not fail the scene either: the stage is run
        again at the fallback scale (gemm.lower_scale), and only a second flag raises.
        hook(codes, cls): called before the completion (the benchmark installs its copy scheduling there)."""
        from .. import _lib, gemm
        for attempt in (0, 1):
            # the GEMM scale THIS attempt runs at: the model (and gemm.SA) is shared by every scene in flight, so
            # another scene may lower it while this one runs
            sa_used = gemm.SA
            codes = self.object_codes(end_points, proposal_features, ids, pc)
            cls = self.cls_codes(end_points, ids)
            # A snapshot of the stream's status word is taken HERE, before the completion, without waiting: a GEMM that
            # overflowed has clipped the codes, and a decoder run on clipped codes could raise ITS range flag -- the
            # completion must then neither lower the decoder's scale for good nor hide the GEMM's flag.
            # The snapshot is read after the completion's own status read has synchronised the stream.
            with torch.cuda.device(pc.device):
                snap = _lib.StatusSnapshot()
            if hook is not None:
                hook(codes, cls)
            try:
                return self.complete(codes, cls, pc.device, return_grids=return_grids, before=snap)
            except _StageFlag as e:
                if e.status & 4 and not e.status & ~4 and attempt == 0:
                    with _lib.BUILD_LOCK:
                        lowered = gemm.lower_scale()
                    # again when the scale came down -- by this scene, or by another scene in flight since this one
                    # started (its run was at the OLD scale: lower_scale() says "already at the fallback" but THIS
                    # scene has never run there); only a flag raised AT the fallback scale is a real overflow
                    if lowered or gemm.SA < sa_used:
                        continue
                _lib.raise_status(e.status)                # FPS abort, or an overflow at the fallback scale
                raise

    def complete(self, codes, cls, device, return_grids=False, before=None):
        """Occupancy completion of the selected proposals + the status read that must follow it.
        FPS exchange time-outs and f16-range flags are reported through the stream's status word, not through
        return codes (the kernels are asynchronous): never hand back results without reading it (waits for THIS
        stream only; several scenes may be in flight, each stream has its own word).  A decoder activation beyond
        the f16 range at the default scale does not fail the scene: the completion is run again at the fallback
        scale (the reference's fp32 decoder cannot overflow, occ_decoder.py:110-123) and only a second flag raises.
        before: a StatusSnapshot taken in front of the completion -- flags in it belong to the stages before (raised as
        _StageFlag for reconstruct() to answer) and are looked at FIRST: the decoder's scale is not touched on their
        account."""
        from .. import _lib
        gen = self.completion.generator
        dec = self.completion.decoder
        run = gen.generate_grids if return_grids else gen.generate_mesh
        ka_used = dec.ka                     # the decoder is shared by the scenes in flight: see reconstruct()
        out = run(codes, cls)
        with torch.cuda.device(device):
            st = _lib.stream_status_bits()
        if before is not None and before.read():
            raise _StageFlag(before.read())
        if st & 2:
            with _lib.BUILD_LOCK:
                lowered = dec.lower_activation_scale()
            if lowered or dec.ka < ka_used:  # this run was folded at a scale that has come down since: once more
                out = run(codes, cls)
                with torch.cuda.device(device):
                    st = (st & ~2) | _lib.stream_status_bits()
        _lib.raise_status(st)
        return out

    @staticmethod
    def check_device_status(device):
        from .. import _lib
        with torch.cuda.device(device):
            _lib.stream_status()
