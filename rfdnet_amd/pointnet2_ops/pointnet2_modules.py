"""Set-abstraction / feature-propagation / STN modules with the reference's
class names, constructor keywords and state_dict keys
(external/pointnet2_ops_lib/pointnet2_ops/pointnet2_modules.py):

  build_shared_mlp          :9-19    Conv2d(1x1, bias=not bn) [+ BatchNorm2d] + ReLU
                                     -> keys mlp_module.{0,1,3,4,6,7}.*
  PointnetSAModuleVotes     :149-260
  PointnetFPModule          :345-405 -> keys mlp.{0,1,3,4}.*
  STN3d / STN_Group         :420-537

The point operators come from the HIP library; the shared MLPs are plain 1x1
convolutions and stay on rocBLAS/MIOpen through torch.
"""
from typing import List

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _ext, pointnet2_utils


def build_shared_mlp(mlp_spec: List[int], bn: bool = True):
    layers = []
    for c_in, c_out in zip(mlp_spec[:-1], mlp_spec[1:]):
        layers.append(nn.Conv2d(c_in, c_out, kernel_size=1, bias=not bn))
        if bn:
            layers.append(nn.BatchNorm2d(c_out))
        layers.append(nn.ReLU(True))
    return nn.Sequential(*layers)


class PointnetSAModuleVotes(nn.Module):
    """Set abstraction with returned sample indices (VoteNet variant)."""

    def __init__(self, *, mlp: List[int], npoint: int = None, radius: float = None,
                 nsample: int = None, bn: bool = True, use_xyz: bool = True,
                 pooling: str = 'max', sigma: float = None, normalize_xyz: bool = False,
                 sample_uniformly: bool = False, ret_unique_cnt: bool = False):
        super().__init__()
        if pooling not in ('max', 'avg', 'rbf'):
            raise ValueError("unknown pooling %r" % pooling)
        self.npoint, self.radius, self.nsample = npoint, radius, nsample
        self.pooling = pooling
        self.use_xyz = use_xyz
        self.sigma = sigma if sigma is not None else (radius / 2 if radius is not None else None)
        self.normalize_xyz = normalize_xyz
        self.ret_unique_cnt = ret_unique_cnt
        if npoint is not None:
            self.grouper = pointnet2_utils.QueryAndGroup(
                radius, nsample, use_xyz=use_xyz, ret_grouped_xyz=True,
                normalize_xyz=normalize_xyz, sample_uniformly=sample_uniformly,
                ret_unique_cnt=ret_unique_cnt)
        else:
            self.grouper = pointnet2_utils.GroupAll(use_xyz, ret_grouped_xyz=True)
        spec = list(mlp)
        if use_xyz and len(spec) > 0:
            spec[0] += 3                                  # xyz channels come first
        self.mlp_module = build_shared_mlp(spec, bn=bn)

    def forward(self, xyz, features=None, inds=None):
        """xyz (B,N,3), features (B,C,N), optional inds (B,npoint) ->
        (new_xyz (B,npoint,3), new_features (B,mlp[-1],npoint), inds)."""
        new_xyz = None
        if self.npoint is not None:
            if inds is None:
                # FPS kernel already holds the winner's coordinates: emit the
                # centres with it instead of transpose + gather + transpose
                inds, new_xyz = _ext.furthest_point_sampling_gather(xyz, self.npoint)
            else:
                assert inds.shape[1] == self.npoint
                new_xyz = torch.gather(xyz, 1, inds.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
        from .. import sa_fused
        if (self.npoint is not None and sa_fused.usable(self.mlp_module, features, self.nsample, self.pooling,
                                                        self.use_xyz)):
            # inference: grouping + shared MLP + max pool in ONE kernel (csrc/sa_fused.hip);
            # the (3+C, npoint, nsample) tensor is never materialised
            idx = pointnet2_utils.ball_query(self.radius, self.nsample, xyz, new_xyz)
            return new_xyz, sa_fused.forward(self.mlp_module, xyz, new_xyz, features, idx, self.radius,
                                             self.normalize_xyz), inds
        grouped_features, grouped_xyz = self.grouper(xyz, new_xyz, features)
        new_features = self.mlp_module(grouped_features)          # (B, C', npoint, nsample)
        if self.pooling == 'max':
            new_features = new_features.max(dim=3)[0]
        elif self.pooling == 'avg':
            new_features = new_features.mean(dim=3)
        else:  # 'rbf'
            rbf = torch.exp(-1 * grouped_xyz.pow(2).sum(1) / (self.sigma ** 2) / 2)
            new_features = torch.sum(new_features * rbf.unsqueeze(1), -1) / float(self.nsample)
        return new_xyz, new_features, inds


class PointnetFPModule(nn.Module):
    """Feature propagation: inverse-distance interpolation from the 3 nearest
    known points, skip concatenation, shared MLP."""

    def __init__(self, mlp, bn=True):
        super().__init__()
        self.mlp = build_shared_mlp(mlp, bn=bn)

    def _fused_layers(self):
        """the shared MLP as [(W, b, relu)] with the eval-mode BatchNorms folded in, or None (no BatchNorm / training)"""
        from ..fold_bn import folded
        mods = list(self.mlp)
        if self.training or len(mods) % 3 or not mods:
            return None
        out = []
        for conv, bn, act in zip(mods[0::3], mods[1::3], mods[2::3]):
            if not (isinstance(conv, nn.Conv2d) and isinstance(bn, nn.BatchNorm2d) and isinstance(act, nn.ReLU)):
                return None
            out.append(folded(conv, bn) + (True,))
        return out

    def forward(self, unknown, known, unknow_feats, known_feats):
        from .. import mlp as fused
        layers = self._fused_layers() if known is not None and known_feats.is_cuda else None
        if layers is not None and 1 <= len(layers) <= 4:
            widths = [known_feats.shape[1] + (0 if unknow_feats is None else unknow_feats.shape[1])] + \
                [W.shape[0] for W, _, _ in layers]
            if known_feats.dtype == torch.float32 and fused.fits(unknown.shape[1], widths):
                # inference: three launches -- three_nn, interpolation with its inverse-distance weights + the skip
                # concatenation, the whole shared MLP -- instead of twenty-two (csrc/interpolate.hip, csrc/mlp_cols.hip)
                dist2, idx = pointnet2_utils._ext.three_nn(unknown.contiguous(), known.contiguous())
                skip = None if unknow_feats is None else unknow_feats.contiguous()
                x = fused.interpolate_cat(known_feats.contiguous(), idx, dist2, skip)
                return fused.mlp_cols(x, layers)
        if known is not None:
            dist, idx = pointnet2_utils.three_nn(unknown, known)
            dist_recip = 1.0 / (dist + 1e-8)
            weight = dist_recip / torch.sum(dist_recip, dim=2, keepdim=True)
            interpolated = pointnet2_utils.three_interpolate(known_feats, idx, weight)
        else:
            interpolated = known_feats.expand(*(list(known_feats.size()[0:2]) + [unknown.size(1)]))
        if unknow_feats is not None:
            new_features = torch.cat([interpolated, unknow_feats], dim=1)
        else:
            new_features = interpolated
        return self.mlp(new_features.unsqueeze(-1)).squeeze(-1)


def weights_init(m):
    """STN layers start from the identity transform (all-zero weights)."""
    if isinstance(m, (nn.Conv2d, nn.Linear)) or m.__class__.__name__.find('Conv2d') != -1:
        if getattr(m, 'weight', None) is not None:
            nn.init.constant_(m.weight.data, 0.0)
        if getattr(m, 'bias', None) is not None:
            nn.init.constant_(m.bias.data, 0.0)


class STN3d(nn.Module):
    """Per-proposal 3x4 affine regressor (PointNet T-net style)."""

    def __init__(self, num_points=2500):
        super().__init__()
        self.num_points = num_points
        self.conv1 = nn.Conv1d(3, 64, 1)
        self.conv2 = nn.Conv1d(64, 128, 1)
        self.conv3 = nn.Conv1d(128, 256, 1)
        self.mp1 = nn.MaxPool1d(num_points)
        self.fc1 = nn.Linear(256, 128)
        self.fc2 = nn.Linear(128, 64)
        self.fc3 = nn.Linear(64, 12)
        self.relu = nn.ReLU(inplace=True)
        self.bn1 = nn.BatchNorm1d(64)
        self.bn2 = nn.BatchNorm1d(128)
        self.bn3 = nn.BatchNorm1d(256)
        self.bn4 = nn.BatchNorm1d(128)
        self.bn5 = nn.BatchNorm1d(64)
        self.apply(weights_init)

    def forward(self, grouped_xyz):
        B, _, P, _ = grouped_xyz.size()
        pts = grouped_xyz.transpose(2, 1).contiguous().view(B * P, 3, self.num_points)
        x = self.relu(self.bn1(self.conv1(pts)))
        x = self.relu(self.bn2(self.conv2(x)))
        x = self.relu(self.bn3(self.conv3(x)))
        x = self.mp1(x).squeeze(2)
        x = self.relu(self.bn4(self.fc1(x)))
        x = self.relu(self.bn5(self.fc2(x)))
        x = self.fc3(x)
        iden = torch.eye(3, 4, device=x.device, dtype=x.dtype).view(1, 12)
        x = (x + iden).view(B * P, 3, 4)
        out = torch.bmm(x[:, :, :3], pts) + x[:, :, 3].unsqueeze(-1)
        return out.view(B, P, 3, -1).transpose(1, 2)


def _stn3d_affine_rows(stn, rows):
    """STN3d's regressor on row-major points: rows (B', P, 3) -> (B', 3, 4).  Eval-mode
    BatchNorms folded into the layers; wide layers on the split-precision GEMM."""
    from ..fold_bn import folded, linear_rows
    from .. import chain
    Bp, P, _ = rows.shape
    x2 = rows.reshape(Bp * P, 3)
    if chain.usable(x2, P, 3):
        # conv1..3 + BatchNorms + ReLU + the max over the group's points as ONE kernel (csrc/pointseg_chain.hip with a
        # 256-wide last layer): the 64- / 128- / 256-wide intermediates of 262 144 points (470 MB) never reach HBM
        g = chain.chain_pool(x2, folded(stn.conv1, stn.bn1), folded(stn.conv2, stn.bn2), folded(stn.conv3, stn.bn3),
                             P, True)
    else:
        h = linear_rows(x2, *folded(stn.conv1, stn.bn1), relu=True)
        h = linear_rows(h, *folded(stn.conv2, stn.bn2), relu=True)
        h = linear_rows(h, *folded(stn.conv3, stn.bn3), relu=True)
        g = h.view(Bp, P, -1).max(dim=1)[0]
    g = linear_rows(g, *folded(stn.fc1, stn.bn4), relu=True)
    g = linear_rows(g, *folded(stn.fc2, stn.bn5), relu=True)
    g = F.linear(g, stn.fc3.weight, stn.fc3.bias)
    g = g + torch.eye(3, 4, device=g.device, dtype=g.dtype).view(1, 12)
    return g.view(Bp, 3, 4)


class STN_Group(nn.Module):
    """Group scan points around box centres, rotate them into the box frame
    (-heading about z) and apply the learned STN3d affine."""

    def __init__(self, radius: float = None, nsample: int = None, use_xyz: bool = True,
                 normalize_xyz: bool = False, sample_uniformly: bool = False,
                 ret_unique_cnt: bool = False):
        super().__init__()
        self.radius, self.nsample = radius, nsample
        self.use_xyz = use_xyz
        self.normalize_xyz = normalize_xyz
        self.ret_unique_cnt = ret_unique_cnt
        self.grouper = pointnet2_utils.QueryAndGroup(
            radius, nsample, use_xyz=use_xyz, ret_grouped_xyz=True, normalize_xyz=normalize_xyz,
            sample_uniformly=sample_uniformly, ret_unique_cnt=ret_unique_cnt)
        self.stn3d = STN3d(num_points=nsample)

    def forward_rows(self, xyz, features=None, new_xyz=None, orientations=None):
        """Inference path that stays ROW-major: -> (points (B*K, P, 3) in the box
        frame after the learned affine, grouped_features (B,C,K,P)).  Same values as
        forward(), without the transposes around every stage."""
        grouped_features, grouped_xyz = self.grouper(xyz, new_xyz, features)
        B, K = orientations.size()
        P = grouped_xyz.shape[3]
        rows = grouped_xyz.permute(0, 2, 3, 1).reshape(B * K, P, 3)
        if rows.is_cuda and rows.dtype == torch.float32 and not torch.is_grad_enabled():
            # the two point transforms as one kernel each (csrc/small_ops.hip) instead of a zero-filled rotation
            # matrix assembled by five indexed stores + bmm, and bmm + add
            from .. import _lib
            rows = rows.contiguous()
            cs = torch.stack([torch.cos(orientations).view(-1), torch.sin(orientations).view(-1)], 1).contiguous()
            rot = torch.empty_like(rows)
            with torch.cuda.device(rows.device):
                _lib.check(_lib.lib().rfd_rows3_rotate_z(B * K, P, rows.data_ptr(), cs.data_ptr(), rot.data_ptr(),
                                                          _lib.current_stream()), "rfd_rows3_rotate_z")
            A = _stn3d_affine_rows(self.stn3d, rot).contiguous()
            out = torch.empty_like(rows)
            with torch.cuda.device(rows.device):
                _lib.check(_lib.lib().rfd_rows3_affine(B * K, P, rot.data_ptr(), A.data_ptr(), out.data_ptr(),
                                                        _lib.current_stream()), "rfd_rows3_affine")
            return out, grouped_features
        cos, sin = torch.cos(orientations).view(-1), torch.sin(orientations).view(-1)
        rot_t = torch.zeros(B * K, 3, 3, device=rows.device, dtype=rows.dtype)   # transpose of the rotation
        rot_t[:, 0, 0] = cos
        rot_t[:, 1, 0] = sin
        rot_t[:, 0, 1] = -sin
        rot_t[:, 1, 1] = cos
        rot_t[:, 2, 2] = 1.
        rows = torch.bmm(rows, rot_t)
        A = _stn3d_affine_rows(self.stn3d, rows)
        rows = torch.bmm(rows, A[:, :, :3].transpose(1, 2)) + A[:, :, 3].unsqueeze(1)
        return rows, grouped_features

    def forward(self, xyz, features=None, new_xyz=None, orientations=None):
        grouped_features, grouped_xyz = self.grouper(xyz, new_xyz, features)
        B, P = orientations.size()
        cos, sin = torch.cos(orientations), torch.sin(orientations)
        rot = torch.zeros(B, P, 3, 3, device=orientations.device, dtype=grouped_xyz.dtype)
        rot[..., 0, 0] = cos
        rot[..., 0, 1] = sin
        rot[..., 1, 0] = -sin
        rot[..., 1, 1] = cos
        rot[..., 2, 2] = 1.
        g = torch.bmm(rot.view(B * P, 3, 3),
                      grouped_xyz.transpose(1, 2).contiguous().view(B * P, 3, -1))
        g = g.view(B, P, 3, -1).transpose(1, 2).contiguous()
        return self.stn3d(g), grouped_features
