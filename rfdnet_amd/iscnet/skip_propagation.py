"""Skip propagation (generation path): group scan points around each predicted
box, align to the box frame, mask with PointSeg and encode to the shape code
(models/iscnet/modules/skip_propagation.py:13-82).  The K x 80 000 ball query
(r = 1 m, 1024 samples), the grouping, fc_pos (csrc/pos_embed.hip) and the wide
layers (csrc/gemm_f16x3.hip) are HIP kernels."""
import torch
import torch.nn.functional as F
from torch import nn

from .. import pos_embed
from ..pointnet2_ops.pointnet2_modules import STN_Group
from .layers import ResnetPointnet
from .pointseg import PointSeg
from .registers import MODULES


@MODULES.register_module
class SkipPropagation(nn.Module):
    def __init__(self, cfg, optim_spec=None):
        super().__init__()
        self.optim_spec = optim_spec
        data = cfg.config['data']
        self.input_feature_dim = int(data['use_color_completion']) * 3 + int(not data['no_height']) * 1
        self.stn = STN_Group(radius=1., nsample=1024, use_xyz=False, normalize_xyz=True)
        self.encoder = ResnetPointnet(c_dim=data['c_dim'], dim=self.input_feature_dim + 3 + 128,
                                      hidden_dim=data['hidden_dim'])
        self.point_seg = PointSeg(num_class=2, channel=self.input_feature_dim + 3)

    def _break_up_pc(self, pc):
        xyz = pc[..., 0:3].contiguous()
        features = (pc[..., 3:3 + self.input_feature_dim].transpose(1, 2).contiguous()
                    if pc.size(-1) > 3 else None)
        return xyz, features

    def generate(self, box_xyz, box_orientations, box_feature, input_point_cloud):
        """box_xyz (B,K,3), box_orientations (B,K), box_feature (B,128,K),
        input_point_cloud (B,N,3+f) -> object codes (B, c_dim, K)."""
        xyz, features = self._break_up_pc(input_point_cloud)
        # the instance-label channel is all zeros at generation time (:52-53)
        features = torch.cat([features, torch.zeros_like(features)], dim=1)
        rows, gfeat = self.stn.forward_rows(xyz, features, box_xyz, box_orientations)   # (B*K,P,3), (B,C,K,P)
        B, _, K, P = gfeat.size()
        inp = torch.cat([rows, gfeat[:, 0].reshape(B * K, P, 1)], dim=2)                  # (B*K, P, 3+f)
        seg_pred, _ = self.point_seg.forward_rows(inp)
        mask = torch.argmax(seg_pred.view(B * K * P, 2), dim=1).view(B * K, P, 1)
        # encoder input = cat([points, box feature repeated over the points]) * mask.
        # The 128 box-feature channels are one vector per proposal, so their share
        # of fc_pos is a per-proposal vector scaled by the per-point 0/1 mask.
        maskf = mask.float()                                                     # (B*K,P,1)
        enc = self.encoder
        d = inp.shape[2]
        box = box_feature.transpose(1, 2).contiguous().view(B * K, -1)          # (B*K,128)
        w = enc.fc_pos.weight
        group = F.linear(box, w[:, d:])                                          # (B*K,2h): once per proposal
        rows = inp.reshape(B * K * P, d)
        if enc.frag_usable(B * K, P) and pos_embed.frag_usable(rows, w, B * K * P, P):
            # the encoder on fragment-ordered split activations: fc_pos writes block 0's input already rectified,
            # scaled and split (the scale exponent is captured ONCE: another scene in flight may lower gemm.SA)
            from .. import gemm
            sa = gemm.SA
            cat, window = enc.frag_input_buffer(B * K, P, rows.device)
            pos_embed.pos_embed_frag(rows, maskf.view(-1), w, enc.fc_pos.bias, group, P, window, sa)
            codes = enc.forward_frag(cat, B * K, P, sa)
            return codes.view(B, K, -1).transpose(1, 2)
        pos = enc.input_buffer(B * K, P, rows.device)                            # (B*K*P,2h) window of block 0's buffer
        if pos_embed.usable(rows, w, pos):
            pos_embed.pos_embed(rows, maskf.view(-1), w, enc.fc_pos.bias, group, P, pos)
        else:
            tmp = F.linear(inp * maskf, w[:, :d], enc.fc_pos.bias)
            pos.copy_(tmp.addcmul_(maskf, group.unsqueeze(1)).view(B * K * P, -1))
        codes = enc.forward_factored(pos.view(B * K, P, -1))
        return codes.view(B, K, -1).transpose(1, 2)
