"""Small fully-connected building blocks used by the skip-propagation encoder
(models/iscnet/modules/layers.py:5-48 ResnetBlockFC, :340-392 ResnetPointnet).
Plain GEMMs: they stay on rocBLAS through torch."""
import torch
import torch.nn as nn


class ResnetBlockFC(nn.Module):
    def __init__(self, size_in, size_out=None, size_h=None):
        super().__init__()
        size_out = size_in if size_out is None else size_out
        size_h = min(size_in, size_out) if size_h is None else size_h
        self.size_in, self.size_h, self.size_out = size_in, size_h, size_out
        self.fc_0 = nn.Linear(size_in, size_h)
        self.fc_1 = nn.Linear(size_h, size_out)
        self.actvn = nn.ReLU(inplace=True)
        self.shortcut = None if size_in == size_out else nn.Linear(size_in, size_out, bias=False)
        nn.init.zeros_(self.fc_1.weight)

    def forward(self, x):
        # NB: the reference's in-place ReLU also rectifies x itself before the
        # shortcut/identity is taken (layers.py:27,38-46) -- reproduced.
        ax = torch.relu(x)
        net = self.fc_0(ax)
        dx = self.fc_1(torch.relu(net))
        x_s = self.shortcut(ax) if self.shortcut is not None else ax
        return x_s + dx


def maxpool(x, dim=-1, keepdim=False):
    return x.max(dim=dim, keepdim=keepdim)[0]


class ResnetPointnet(nn.Module):
    """PointNet encoder with 5 ResNet blocks and max-pool context (c_dim out)."""

    def __init__(self, c_dim=128, dim=3, hidden_dim=128):
        super().__init__()
        self.c_dim = c_dim
        self.fc_pos = nn.Linear(dim, 2 * hidden_dim)
        for i in range(5):
            setattr(self, 'block_%d' % i, ResnetBlockFC(2 * hidden_dim, hidden_dim))
        self.fc_c = nn.Linear(hidden_dim, c_dim)
        self.actvn = nn.ReLU()
        self.pool = maxpool

    def forward(self, p):
        net = self.block_0(self.fc_pos(p))
        for i in range(1, 5):
            pooled = self.pool(net, dim=1, keepdim=True).expand(net.size())
            net = getattr(self, 'block_%d' % i)(torch.cat([net, pooled], dim=2))
        net = self.pool(net, dim=1)
        return self.fc_c(self.actvn(net))

    def forward_factored(self, pos_term):
        """Same function as forward(), evaluated without ever materialising
        cat([net, pooled]): the pooled half of every block input is constant over
        a proposal's points, so W[:, 512:] . relu(pooled) is ONE vector per
        proposal and only the 512-wide per-point half goes through the big GEMMs
        (-40 % FLOPs in blocks 1-4).  fc_0 and the shortcut share their input, so
        their per-point halves run as one GEMM.  `pos_term` = fc_pos output
        (B,T,2*hidden), passed in because its own input is factored by the caller."""
        import torch.nn.functional as F
        h = self.block_0.size_h
        net = self.block_0(pos_term)
        for i in range(1, 5):
            blk = getattr(self, 'block_%d' % i)
            pooled = torch.relu(self.pool(net, dim=1))                       # (B,h)
            a = torch.relu(net)                                              # (B,T,h)
            w_pt = torch.cat([blk.fc_0.weight[:, :h], blk.shortcut.weight[:, :h]], 0)      # (2h,h)
            w_pl = torch.cat([blk.fc_0.weight[:, h:], blk.shortcut.weight[:, h:]], 0)
            bias = torch.cat([blk.fc_0.bias, torch.zeros_like(blk.fc_0.bias)])
            both = F.linear(a, w_pt) + F.linear(pooled, w_pl, bias).unsqueeze(1)           # (B,T,2h)
            dx = blk.fc_1(torch.relu(both[..., :h]))
            net = both[..., h:] + dx
        net = self.pool(net, dim=1)
        return self.fc_c(self.actvn(net))
