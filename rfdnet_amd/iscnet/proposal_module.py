"""Vote aggregation + proposal head (models/iscnet/modules/proposal_module.py:13-124).

`seed_fps` (the shipped setting): FPS over the SEED positions picks `num_target` cluster centres,
one set-abstraction layer (ball query r = 0.3, 16 neighbours) aggregates the VOTES around them,
and a three-layer 1x1-convolution head emits 2 + 3 + 2*NH + 4*NS + NC = 69 channels per proposal.
Attribute names (vote_aggregation, conv1..3, bn1..2) follow the reference for checkpoint
parity; the channel layout of the head is described once, as a table."""
from collections import OrderedDict

import torch
from torch import nn

from ..pointnet2_ops import pointnet2_utils
from ..pointnet2_ops.pointnet2_modules import PointnetSAModuleVotes
from .registers import MODULES
from .vote_module import _head


def head_layout(num_heading_bin, num_size_cluster, num_class):
    """name -> (first channel, width) of the head output, in the reference's order (:13-39)."""
    fields = (('objectness_scores', 2), ('center_offset', 3), ('heading_scores', num_heading_bin),
              ('heading_residuals_normalized', num_heading_bin), ('size_scores', num_size_cluster),
              ('size_residuals_normalized', 3 * num_size_cluster), ('sem_cls_scores', num_class))
    layout, at = OrderedDict(), 0
    for name, width in fields:
        layout[name] = (at, width)
        at += width
    return layout, at


def decode_scores(net, end_points, num_heading_bin, num_size_cluster):
    """Slice the (B, C, P) head output into the named predictions the rest of the pipeline reads."""
    per_prop = net.transpose(2, 1)                                        # (B,P,C)
    B, P, C = per_prop.shape
    layout, _ = head_layout(num_heading_bin, num_size_cluster, C - (5 + 2 * num_heading_bin + 4 * num_size_cluster))
    for name, (at, width) in layout.items():
        piece = per_prop[:, :, at:at + width]
        if name == 'center_offset':
            end_points['center'] = end_points['aggregated_vote_xyz'] + piece
        elif name == 'size_residuals_normalized':
            end_points[name] = piece.view(B, P, num_size_cluster, 3)
        else:
            end_points[name] = piece
    return end_points


@MODULES.register_module
class ProposalModule(nn.Module):
    def __init__(self, cfg, optim_spec=None):
        super().__init__()
        self.optim_spec = optim_spec
        self.cfg = cfg
        ds, data = cfg.dataset_config, cfg.config['data']
        self.num_class, self.num_heading_bin = ds.num_class, ds.num_heading_bin
        self.num_size_cluster, self.mean_size_arr = ds.num_size_cluster, ds.mean_size_arr
        self.num_proposal = data['num_target']
        self.sampling = data['cluster_sampling']
        self.seed_feat_dim = 256
        self.vote_aggregation = PointnetSAModuleVotes(
            npoint=self.num_proposal, radius=0.3, nsample=16, mlp=[self.seed_feat_dim, 128, 128, 128],
            use_xyz=True, normalize_xyz=True)
        _, out_ch = head_layout(self.num_heading_bin, self.num_size_cluster, self.num_class)
        _head(self, (128, 128, out_ch), 128)

    def _cluster_indices(self, end_points, device):
        """Which seeds become cluster centres (None: let the aggregation layer run FPS on the votes)."""
        seeds = end_points['seed_xyz']
        if self.sampling == 'seed_fps':
            return pointnet2_utils.furthest_point_sample(seeds, self.num_proposal)
        if self.sampling == 'random':
            return torch.randint(0, seeds.shape[1], (seeds.shape[0], self.num_proposal), dtype=torch.int,
                                 device=device)
        if self.sampling == 'vote_fps':
            return None
        raise ValueError('Unknown sampling strategy: %s' % self.sampling)

    def _head_fused(self, x):
        """the 69-channel head (conv1-bn1-ReLU-conv2-bn2-ReLU-conv3, proposal_module.py:85-124) as ONE kernel at
        inference (csrc/mlp_cols.hip, BatchNorms folded); None when that path does not apply"""
        from .. import mlp as fused
        from ..fold_bn import folded
        if self.training or not x.is_cuda:
            return None
        layers = [folded(self.conv1, self.bn1) + (True,), folded(self.conv2, self.bn2) + (True,),
                  folded(self.conv3) + (False,)]
        x = x.contiguous()
        if not fused.usable(x, [x.shape[1]] + [W.shape[0] for W, _, _ in layers]):
            return None
        return fused.mlp_cols(x, layers)

    def forward(self, xyz, features, end_points, export_proposal_feature=False):
        """xyz (B,K,3) vote positions, features (B,C,K) vote features -> end_points with the
        decoded head (+ the 128-d proposal features on request)."""
        picked = self._cluster_indices(end_points, xyz.device)
        xyz, features, fps_inds = self.vote_aggregation(xyz, features, picked)
        end_points['aggregated_vote_xyz'] = xyz
        end_points['aggregated_vote_inds'] = fps_inds if picked is None else picked
        scores = self._head_fused(features)
        if scores is None:
            h = features
            for conv, bn in ((self.conv1, self.bn1), (self.conv2, self.bn2)):
                h = torch.relu(bn(conv(h)))
            scores = self.conv3(h)
        end_points = decode_scores(scores, end_points, self.num_heading_bin, self.num_size_cluster)
        return end_points, (features if export_proposal_feature else None)
