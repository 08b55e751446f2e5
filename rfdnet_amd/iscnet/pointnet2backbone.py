"""PointNet++ backbone of VoteNet / RfD-Net: 4 set-abstraction + 2
feature-propagation layers (models/iscnet/modules/pointnet2backbone.py:11-125).
Same sub-module names (sa1..sa4, fp1, fp2) and end_points keys."""
import torch.nn as nn

from ..pointnet2_ops.pointnet2_modules import PointnetFPModule, PointnetSAModuleVotes
from .registers import MODULES

# (npoint, radius, nsample, mlp widths) -- pointnet2backbone.py:27-61
SA_SPECS = (
    (2048, 0.2, 64, (64, 64, 128)),
    (1024, 0.4, 32, (128, 128, 256)),
    (512, 0.8, 16, (128, 128, 256)),
    (256, 1.2, 16, (128, 128, 256)),
)


@MODULES.register_module
class Pointnet2Backbone(nn.Module):
    def __init__(self, cfg, optim_spec=None):
        super().__init__()
        self.optim_spec = optim_spec
        data = cfg.config['data']
        self.input_feature_dim = int(data['use_color_detection']) * 3 + int(not data['no_height']) * 1
        c_in = self.input_feature_dim
        for i, (npoint, radius, nsample, widths) in enumerate(SA_SPECS, start=1):
            setattr(self, 'sa%d' % i, PointnetSAModuleVotes(
                npoint=npoint, radius=radius, nsample=nsample, mlp=[c_in] + list(widths),
                use_xyz=True, normalize_xyz=True))
            c_in = widths[-1]
        self.fp1 = PointnetFPModule(mlp=[256 + 256, 256, 256])
        self.fp2 = PointnetFPModule(mlp=[256 + 256, 256, 256])

    def _break_up_pc(self, pc):
        xyz = pc[..., 0:3].contiguous()
        features = (pc[..., 3:3 + self.input_feature_dim].transpose(1, 2).contiguous()
                    if pc.size(-1) > 3 else None)
        return xyz, features

    def forward(self, pointcloud, end_points=None):
        """pointcloud (B, N, 3 + input_feature_dim) -> end_points dict."""
        end_points = end_points if end_points else {}
        xyz, features = self._break_up_pc(pointcloud)
        for i in (1, 2, 3, 4):
            xyz, features, fps_inds = getattr(self, 'sa%d' % i)(xyz, features)
            if i <= 2:
                end_points['sa%d_inds' % i] = fps_inds
            end_points['sa%d_xyz' % i] = xyz
            end_points['sa%d_features' % i] = features
        features = self.fp1(end_points['sa3_xyz'], end_points['sa4_xyz'],
                            end_points['sa3_features'], end_points['sa4_features'])
        features = self.fp2(end_points['sa2_xyz'], end_points['sa3_xyz'],
                            end_points['sa2_features'], features)
        end_points['fp2_features'] = features
        end_points['fp2_xyz'] = end_points['sa2_xyz']
        num_seed = end_points['fp2_xyz'].shape[1]
        # FPS of an FPS-ordered set returns 0..n-1, so the seeds are the first
        # num_seed of sa1's samples (pointnet2backbone.py:104,124)
        end_points['fp2_inds'] = end_points['sa1_inds'][:, 0:num_seed]
        return end_points
