"""PointNet segmentation net that masks the points of each proposal
(models/iscnet/modules/pointseg.py).  Sub-module and parameter names follow the
reference so its checkpoints load; dense 1x1 convolutions stay on rocBLAS."""
import torch
import torch.nn.functional as F
from torch import nn


class _TNet(nn.Module):
    """Shared trunk of STN3d (:7-42) and STNkd (:45-79): k_in -> k_out*k_out."""

    def __init__(self, k_in, k_out):
        super().__init__()
        self.conv1 = nn.Conv1d(k_in, 64, 1)
        self.conv2 = nn.Conv1d(64, 128, 1)
        self.conv3 = nn.Conv1d(128, 1024, 1)
        self.fc1 = nn.Linear(1024, 512)
        self.fc2 = nn.Linear(512, 256)
        self.fc3 = nn.Linear(256, k_out * k_out)
        self.relu = nn.ReLU()
        self.bn1 = nn.BatchNorm1d(64)
        self.bn2 = nn.BatchNorm1d(128)
        self.bn3 = nn.BatchNorm1d(1024)
        self.bn4 = nn.BatchNorm1d(512)
        self.bn5 = nn.BatchNorm1d(256)
        self._k_out = k_out

    def forward(self, x):
        x = F.relu(self.bn1(self.conv1(x)))
        x = F.relu(self.bn2(self.conv2(x)))
        x = F.relu(self.bn3(self.conv3(x)))
        x = torch.max(x, 2)[0]
        x = F.relu(self.bn4(self.fc1(x)))
        x = F.relu(self.bn5(self.fc2(x)))
        x = self.fc3(x)
        k = self._k_out
        x = x + torch.eye(k, device=x.device, dtype=x.dtype).view(1, k * k)
        return x.view(-1, k, k)


    def forward_rows(self, x, P):
        """Row-major inference path: x (B*P, k_in) -> (B, k_out, k_out).  BN folded,
        the wide layers on the split-precision GEMM with fused bias + ReLU."""
        from ..fold_bn import folded, linear_rows, linear_rows_pooled
        from .. import chain
        if chain.usable(x, P, x.shape[1]) and x.shape[1] in (64,) + tuple(range(1, 9)):
            # the whole conv1 -> conv2 -> conv3 -> max chain in one kernel (csrc/pointseg_chain.hip)
            g = chain.chain_pool(x, folded(self.conv1, self.bn1), folded(self.conv2, self.bn2),
                                 folded(self.conv3, self.bn3), P, relu3=True)
        else:
            h = linear_rows(x, *folded(self.conv1, self.bn1), relu=True)
            h = linear_rows(h, *folded(self.conv2, self.bn2), relu=True)
            g = linear_rows_pooled(h, *folded(self.conv3, self.bn3), rows_per_group=P)    # relu + max over the points
        g = linear_rows(g, *folded(self.fc1, self.bn4), relu=True)
        g = linear_rows(g, *folded(self.fc2, self.bn5), relu=True)
        g = F.linear(g, self.fc3.weight, self.fc3.bias)
        k = self._k_out
        g = g + torch.eye(k, device=g.device, dtype=g.dtype).view(1, k * k)
        return g.view(-1, k, k)


class STN3d(_TNet):
    def __init__(self, channel):
        super().__init__(channel, 3)


class STNkd(_TNet):
    def __init__(self, k=64):
        super().__init__(k, k)
        self.k = k


class PointNetEncoder(nn.Module):
    def __init__(self, global_feat=True, feature_transform=False, channel=3):
        super().__init__()
        self.stn = STN3d(channel)
        self.conv1 = nn.Conv1d(channel, 64, 1)
        self.conv2 = nn.Conv1d(64, 128, 1)
        self.conv3 = nn.Conv1d(128, 1024, 1)
        self.bn1 = nn.BatchNorm1d(64)
        self.bn2 = nn.BatchNorm1d(128)
        self.bn3 = nn.BatchNorm1d(1024)
        self.global_feat = global_feat
        self.feature_transform = feature_transform
        if feature_transform:
            self.fstn = STNkd(k=64)

    def forward_with_pointfeat(self, x):
        """(global feature (B,1024), trans, trans_feat, per-point features (B,64,N))"""
        B, D, N = x.size()
        trans = self.stn(x)
        x = x.transpose(2, 1)
        if D > 3:
            x = torch.cat([torch.bmm(x[..., :3], trans), x[..., 3:]], dim=2)
        else:
            x = torch.bmm(x, trans)
        x = F.relu(self.bn1(self.conv1(x.transpose(2, 1))))
        trans_feat = None
        if self.feature_transform:
            trans_feat = self.fstn(x)
            x = torch.bmm(x.transpose(2, 1), trans_feat).transpose(2, 1)
        pointfeat = x
        x = F.relu(self.bn2(self.conv2(x)))
        x = self.bn3(self.conv3(x))
        x = torch.max(x, 2)[0]
        return x, trans, trans_feat, pointfeat

    def forward(self, x):
        B, D, N = x.size()
        trans = self.stn(x)
        x = x.transpose(2, 1)
        if D > 3:
            x = torch.cat([torch.bmm(x[..., :3], trans), x[..., 3:]], dim=2)
        else:
            x = torch.bmm(x, trans)
        x = F.relu(self.bn1(self.conv1(x.transpose(2, 1))))
        trans_feat = None
        if self.feature_transform:
            trans_feat = self.fstn(x)
            x = torch.bmm(x.transpose(2, 1), trans_feat).transpose(2, 1)
        pointfeat = x
        x = F.relu(self.bn2(self.conv2(x)))
        x = self.bn3(self.conv3(x))
        x = torch.max(x, 2, keepdim=True)[0].view(-1, 1024)
        if self.global_feat:
            return x, trans, trans_feat
        x = x.view(-1, 1024, 1).repeat(1, 1, N)
        return torch.cat([x, pointfeat], 1), trans, trans_feat


class PointSeg(nn.Module):
    def __init__(self, num_class, channel):
        super().__init__()
        self.k = num_class
        self.feat = PointNetEncoder(global_feat=False, feature_transform=True, channel=channel)
        self.conv1 = nn.Conv1d(1088, 512, 1)
        self.conv2 = nn.Conv1d(512, 256, 1)
        self.conv3 = nn.Conv1d(256, 128, 1)
        self.conv4 = nn.Conv1d(128, self.k, 1)
        self.bn1 = nn.BatchNorm1d(512)
        self.bn2 = nn.BatchNorm1d(256)
        self.bn3 = nn.BatchNorm1d(128)

    def forward(self, x):
        B, _, n_pts = x.size()
        if not self.training:
            return self._forward_factored(x)
        x, _, trans_feat = self.feat(x)
        x = F.relu(self.bn1(self.conv1(x)))
        x = F.relu(self.bn2(self.conv2(x)))
        x = F.relu(self.bn3(self.conv3(x)))
        x = self.conv4(x).transpose(2, 1).contiguous()
        x = F.log_softmax(x.view(-1, self.k), dim=-1).view(B, n_pts, self.k)
        return x, trans_feat

    def forward_rows(self, inp):
        """Inference path on ROW-major points: inp (B, P, D) with xyz first ->
        (log-probabilities (B,P,k), trans_feat).  Same function as forward() on
        inp.transpose(1, 2); BatchNorms folded into the layers, the global-feature
        share of the head's first layer reduced to one vector per proposal, wide
        layers on the split-precision GEMM."""
        from ..fold_bn import folded, linear_rows, linear_rows_pooled
        B, P, D = inp.shape
        enc = self.feat
        x = inp.reshape(B * P, D)
        trans = enc.stn.forward_rows(x, P)                                   # (B,3,3)
        xyz = torch.bmm(inp[..., :3], trans)
        x = (torch.cat([xyz, inp[..., 3:]], dim=2) if D > 3 else xyz).reshape(B * P, D)
        h = linear_rows(x, *folded(enc.conv1, enc.bn1), relu=True)           # (M,64)
        trans_feat = None
        if enc.feature_transform:
            trans_feat = enc.fstn.forward_rows(h, P)                         # (B,64,64)
            h = torch.bmm(h.view(B, P, -1), trans_feat).reshape(B * P, -1)
        pointfeat = h
        from .. import chain
        if chain.usable(pointfeat, P, pointfeat.shape[1]) and pointfeat.shape[1] == 64:
            g = chain.chain_pool(pointfeat, None, folded(enc.conv2, enc.bn2), folded(enc.conv3, enc.bn3), P, relu3=False)
        else:
            h = linear_rows(pointfeat, *folded(enc.conv2, enc.bn2), relu=True)
            g = linear_rows_pooled(h, *folded(enc.conv3, enc.bn3), P, relu=False)   # (B,1024), product never written
        # head conv1 on cat([global (1024, per proposal), pointfeat (64, per point)]) + bn1
        W, b = folded(self.conv1, self.bn1)
        c = self.__dict__.get('_head_split')
        if c is None or c[0] is not W:                       # per-point / per-proposal column halves
            from .. import _lib
            with _lib.BUILD_LOCK:      # shared across host threads: built once, published before it is stored
                c = self.__dict__.get('_head_split')
                if c is None or c[0] is not W:
                    c = (W, W[:, 1024:].contiguous(), W[:, :1024].contiguous(), torch.zeros_like(b))
                    _lib.publish(W.device)
                    self.__dict__['_head_split'] = c
        gbias = F.linear(g, c[2], b).contiguous()                              # (B,512): conv1's global-feature share + bias
        if chain.head_usable(pointfeat, P, self.k):
            # conv1 (point-feature columns) -> conv2 -> conv3 -> conv4 in one kernel (csrc/pointseg_chain.hip)
            y = chain.head_scores(pointfeat, P, c[1], gbias, folded(self.conv2, self.bn2), folded(self.conv3, self.bn3),
                                  self.conv4.weight[:, :, 0], self.conv4.bias)
        else:
            y = linear_rows(pointfeat, c[1], c[3], relu=True, gbias=gbias, rows_per_group=P)
            y = linear_rows(y, *folded(self.conv2, self.bn2), relu=True)
            y = linear_rows(y, *folded(self.conv3, self.bn3), relu=True)
            y = F.linear(y, self.conv4.weight[:, :, 0], self.conv4.bias)
        return F.log_softmax(y, dim=-1).view(B, P, self.k), trans_feat

    def _forward_factored(self, x):
        """Inference path: the 1024 global-feature channels of the 1088-channel
        head input are identical for all points of a proposal, so their share of
        conv1 is one vector per proposal; only the 64 point-feature channels go
        through the per-point GEMM (conv1: 1088x512 -> 64x512 MACs per point)."""
        B, _, n_pts = x.size()
        enc = self.feat
        gf = enc.global_feat
        enc.global_feat = True                      # ask the encoder for the un-tiled global vector
        try:
            g, _, trans_feat, pointfeat = enc.forward_with_pointfeat(x)
        finally:
            enc.global_feat = gf
        w = self.conv1.weight[:, :, 0]
        head = F.linear(g, w[:, :1024], self.conv1.bias)                       # (B,512)
        y = F.conv1d(pointfeat, w[:, 1024:].unsqueeze(-1)) + head.unsqueeze(-1)
        y = F.relu(self.bn1(y))
        y = F.relu(self.bn2(self.conv2(y)))
        y = F.relu(self.bn3(self.conv3(y)))
        y = self.conv4(y).transpose(2, 1).contiguous()
        y = F.log_softmax(y.view(-1, self.k), dim=-1).view(B, n_pts, self.k)
        return y, trans_feat
