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
        from .. import gemm, resblock
        h = self.block_0.size_h
        B, T, _ = pos_term.shape
        x2 = pos_term.reshape(B * T, 2 * h)
        if h == resblock.HIDDEN and resblock.usable(x2, T):
            # one fused kernel per block (csrc/resblock.hip): fc_0, ReLU, fc_1 and the
            # shortcut without any intermediate leaving the registers
            blk = self.block_0
            zero = torch.zeros(1, h, device=x2.device, dtype=x2.dtype)
            net = resblock.forward(blk, x2, (zero + blk.fc_0.bias).contiguous(),
                                   (zero + blk.fc_1.bias).contiguous(), B * T)
            for i in range(1, 5):
                blk = getattr(self, 'block_%d' % i)
                pooled = torch.relu(self.pool(net.view(B, T, h), dim=1))                    # (B,h)
                g0 = F.linear(pooled, blk.fc_0.weight[:, h:], blk.fc_0.bias)                # once per proposal
                gs = F.linear(pooled, blk.shortcut.weight[:, h:], blk.fc_1.bias)
                net = resblock.forward(blk, net, g0, gs, T)
            net = self.pool(net.view(B, T, h), dim=1)
            return self.fc_c(self.actvn(net))
        fast = gemm.usable(B * T, 2 * h, 2 * h, pos_term.view(B * T, -1)) and h % 128 == 0

        def stacked(i):
            """[fc_0 ; shortcut] weights of block i, split into the per-point and the
            pooled column halves (cached per parameter version)."""
            blk = getattr(self, 'block_%d' % i)
            key = (i, blk.fc_0.weight._version, blk.shortcut.weight._version, blk.fc_0.bias._version,
                   blk.fc_0.weight.data_ptr())
            c = self.__dict__.setdefault('_stack_cache', {})
            if c.get(i, (None,))[0] != key:
                w = torch.cat([blk.fc_0.weight, blk.shortcut.weight], 0).detach()
                bias = torch.cat([blk.fc_0.bias, torch.zeros_like(blk.fc_0.bias)]).detach()
                c[i] = (key, w[:, :h].contiguous(), w[:, h:].contiguous(), bias, w.contiguous())
            return blk, c[i]

        # block 0: both halves of its 2h-wide input are per-point
        blk, (_, _, _, bias0, w_full) = stacked(0)
        # the max-pool over a proposal's points (+ the ReLU every consumer applies to it) rides in
        # the epilogue of each block's second GEMM
        fuse_pool = fast and gemm.pool_usable(B * T, h, h, T)

        def second(blk, both, last=False):
            pooled = torch.zeros(B, h, device=both.device, dtype=both.dtype) if fuse_pool else None
            out = gemm.linear(both[:, :h], blk.fc_1.weight, bias=blk.fc_1.bias, residual=both[:, h:],
                              relu_in=True, rows_per_group=T, pool=pooled,
                              store=not (last and fuse_pool))      # the last block is only pooled
            return (out.view(B, T, h) if out is not None else None), pooled

        pooled = None
        if fast:
            x2 = pos_term.view(B * T, 2 * h)
            both = gemm.linear(x2, w_full, bias=bias0, relu_in=True)                       # (M,2h)
            net, pooled = second(blk, both)
        else:
            net = self.block_0(pos_term)
        for i in range(1, 5):
            blk, (_, w_pt, w_pl, bias, _) = stacked(i)
            if pooled is None:
                pooled = torch.relu(self.pool(net, dim=1))                   # (B,h)
            gb = F.linear(pooled, w_pl, bias)                                # (B,2h): once per proposal
            pooled = None
            if fast:
                both = gemm.linear(net.view(B * T, h), w_pt, gbias=gb, rows_per_group=T, relu_in=True)
                net, pooled = second(blk, both, last=(i == 4))
            else:
                both = F.linear(torch.relu(net), w_pt) + gb.unsqueeze(1)     # (B,T,2h)
                dx = blk.fc_1(torch.relu(both[..., :h]))
                net = both[..., h:] + dx
        if pooled is not None:                                               # = relu(max over the points)
            return self.fc_c(pooled)
        net = self.pool(net, dim=1)
        return self.fc_c(self.actvn(net))
