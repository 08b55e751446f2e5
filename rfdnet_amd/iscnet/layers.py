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
