"""Hough voting head (models/iscnet/modules/vote_module.py:12-61): three 1x1
convolutions produce a 3-D offset and a feature residual per seed."""
import torch
import torch.nn.functional as F

from .registers import MODULES


@MODULES.register_module
class VotingModule(torch.nn.Module):
    def __init__(self, cfg, optim_spec=None):
        super().__init__()
        self.optim_spec = optim_spec
        self.vote_factor = cfg.config['data']['vote_factor']
        self.in_dim = 256
        self.out_dim = self.in_dim            # residual features: in == out
        self.conv1 = torch.nn.Conv1d(self.in_dim, self.in_dim, 1)
        self.conv2 = torch.nn.Conv1d(self.in_dim, self.in_dim, 1)
        self.conv3 = torch.nn.Conv1d(self.in_dim, (3 + self.out_dim) * self.vote_factor, 1)
        self.bn1 = torch.nn.BatchNorm1d(self.in_dim)
        self.bn2 = torch.nn.BatchNorm1d(self.in_dim)

    def forward(self, seed_xyz, seed_features):
        """seed_xyz (B,S,3), seed_features (B,256,S) ->
        vote_xyz (B,S*vf,3), vote_features (B,256,S*vf)."""
        B, S = seed_xyz.shape[0], seed_xyz.shape[1]
        vf = self.vote_factor
        net = F.relu(self.bn1(self.conv1(seed_features)))
        net = F.relu(self.bn2(self.conv2(net)))
        net = self.conv3(net).transpose(2, 1).view(B, S, vf, 3 + self.out_dim)
        vote_xyz = (seed_xyz.unsqueeze(2) + net[..., 0:3]).contiguous().view(B, S * vf, 3)
        vote_features = seed_features.transpose(2, 1).unsqueeze(2) + net[..., 3:]
        vote_features = vote_features.contiguous().view(B, S * vf, self.out_dim)
        return vote_xyz, vote_features.transpose(2, 1).contiguous()
