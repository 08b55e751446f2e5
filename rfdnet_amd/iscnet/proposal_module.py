"""Vote aggregation + proposal head (models/iscnet/modules/proposal_module.py:
13-124).  `seed_fps`: FPS on the SEED positions picks num_target cluster
centres; the votes are then aggregated around them by one set-abstraction
layer (ball query r=0.3, 16 samples) and a 3-layer 1x1-conv head emits
2 + 3 + 2*NH + 4*NS + NC = 69 channels per proposal."""
import torch
import torch.nn.functional as F

from ..pointnet2_ops import pointnet2_utils
from ..pointnet2_ops.pointnet2_modules import PointnetSAModuleVotes
from .registers import MODULES


def decode_scores(net, end_points, num_heading_bin, num_size_cluster):
    """Slice the (B, 69, P) head output into named predictions (:13-39)."""
    t = net.transpose(2, 1)
    B, P = t.shape[0], t.shape[1]
    nh, ns = num_heading_bin, num_size_cluster
    end_points['objectness_scores'] = t[:, :, 0:2]
    end_points['center'] = end_points['aggregated_vote_xyz'] + t[:, :, 2:5]
    end_points['heading_scores'] = t[:, :, 5:5 + nh]
    end_points['heading_residuals_normalized'] = t[:, :, 5 + nh:5 + 2 * nh]
    end_points['size_scores'] = t[:, :, 5 + 2 * nh:5 + 2 * nh + ns]
    end_points['size_residuals_normalized'] = \
        t[:, :, 5 + 2 * nh + ns:5 + 2 * nh + 4 * ns].view([B, P, ns, 3])
    end_points['sem_cls_scores'] = t[:, :, 5 + 2 * nh + 4 * ns:]
    return end_points


@MODULES.register_module
class ProposalModule(torch.nn.Module):
    def __init__(self, cfg, optim_spec=None):
        super().__init__()
        self.optim_spec = optim_spec
        self.cfg = cfg
        dc = cfg.dataset_config
        self.num_class = dc.num_class
        self.num_heading_bin = dc.num_heading_bin
        self.num_size_cluster = dc.num_size_cluster
        self.mean_size_arr = dc.mean_size_arr
        self.num_proposal = cfg.config['data']['num_target']
        self.sampling = cfg.config['data']['cluster_sampling']
        self.seed_feat_dim = 256
        self.vote_aggregation = PointnetSAModuleVotes(
            npoint=self.num_proposal, radius=0.3, nsample=16,
            mlp=[self.seed_feat_dim, 128, 128, 128], use_xyz=True, normalize_xyz=True)
        out_ch = 2 + 3 + self.num_heading_bin * 2 + self.num_size_cluster * 4 + self.num_class
        self.conv1 = torch.nn.Conv1d(128, 128, 1)
        self.conv2 = torch.nn.Conv1d(128, 128, 1)
        self.conv3 = torch.nn.Conv1d(128, out_ch, 1)
        self.bn1 = torch.nn.BatchNorm1d(128)
        self.bn2 = torch.nn.BatchNorm1d(128)

    def forward(self, xyz, features, end_points, export_proposal_feature=False):
        """xyz (B,K,3) vote positions, features (B,C,K) vote features."""
        if self.sampling == 'vote_fps':
            xyz, features, sample_inds = self.vote_aggregation(xyz, features)
        elif self.sampling == 'seed_fps':
            sample_inds = pointnet2_utils.furthest_point_sample(end_points['seed_xyz'], self.num_proposal)
            xyz, features, _ = self.vote_aggregation(xyz, features, sample_inds)
        elif self.sampling == 'random':
            num_seed = end_points['seed_xyz'].shape[1]
            sample_inds = torch.randint(0, num_seed, (end_points['seed_xyz'].shape[0], self.num_proposal),
                                        dtype=torch.int, device=xyz.device)
            xyz, features, _ = self.vote_aggregation(xyz, features, sample_inds)
        else:
            raise ValueError('Unknown sampling strategy: %s' % self.sampling)
        end_points['aggregated_vote_xyz'] = xyz
        end_points['aggregated_vote_inds'] = sample_inds
        net = F.relu(self.bn1(self.conv1(features)))
        net = F.relu(self.bn2(self.conv2(net)))
        net = self.conv3(net)
        end_points = decode_scores(net, end_points, self.num_heading_bin, self.num_size_cluster)
        return end_points, (features if export_proposal_feature else None)
