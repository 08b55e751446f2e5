"""Hough voting head: every seed casts `vote_factor` votes = its position plus a learnt offset,
carrying its feature plus a learnt residual (models/iscnet/modules/vote_module.py:12-61).

Parameter names (conv1..3, bn1..2) are the reference's, so its checkpoints load.  The head's
last convolution emits, per vote v, the block [offset(3) | residual(256)] at channels
v*(3+256)..; the outputs are assembled channel-major directly, without the reference's two
transposes of the (B, S, vf, 259) tensor."""
import torch
from torch import nn

from .registers import MODULES

SEED_FEATURE_DIM = 256


def _head(module, widths, d_in):
    """conv{i} for every layer, then bn{i} for all but the last, as attributes of `module`
    (registration order = the reference's state_dict order: all convolutions first)."""
    for i, w in enumerate(widths, start=1):
        setattr(module, 'conv%d' % i, nn.Conv1d(d_in, w, kernel_size=1))
        d_in = w
    for i, w in enumerate(widths[:-1], start=1):
        setattr(module, 'bn%d' % i, nn.BatchNorm1d(w))


@MODULES.register_module
class VotingModule(nn.Module):
    def __init__(self, cfg, optim_spec=None):
        super().__init__()
        self.optim_spec = optim_spec                      # only read by the (out-of-scope) trainer
        self.vote_factor = int(cfg.config['data']['vote_factor'])
        d = SEED_FEATURE_DIM
        self.in_dim = self.out_dim = d                     # residual connection: widths must agree
        _head(self, (d, d, (3 + d) * self.vote_factor), d)

    def _head_fused(self, x):
        """conv1-bn1-ReLU-conv2-bn2-ReLU-conv3 as ONE kernel at inference (csrc/mlp_cols.hip; BatchNorms folded),
        None when that path does not apply (training, CPU, a seed count that is no multiple of 8)"""
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

    def forward(self, seed_xyz, seed_features):
        """seed_xyz (B,S,3), seed_features (B,256,S) -> vote_xyz (B,S*vf,3), vote_features
        (B,256,S*vf); vote s*vf + v belongs to seed s."""
        B, S, _ = seed_xyz.shape
        vf, d = self.vote_factor, self.out_dim
        out = self._head_fused(seed_features)
        if out is None:
            h = seed_features
            for conv, bn in ((self.conv1, self.bn1), (self.conv2, self.bn2)):
                h = torch.relu(bn(conv(h)))
            out = self.conv3(h)
        out = out.view(B, vf, 3 + d, S)                                  # [vote][offset | residual][seed]
        offsets = out[:, :, :3].permute(0, 3, 1, 2)                      # (B,S,vf,3)
        vote_xyz = (seed_xyz.unsqueeze(2) + offsets).reshape(B, S * vf, 3)
        feats = seed_features.unsqueeze(1) + out[:, :, 3:]               # (B,vf,d,S)
        vote_features = feats.permute(0, 2, 3, 1).reshape(B, d, S * vf)
        return vote_xyz, vote_features
