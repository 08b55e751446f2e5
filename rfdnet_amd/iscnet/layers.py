"""Small fully-connected building blocks used by the skip-propagation encoder
(models/iscnet/modules/layers.py:5-48 ResnetBlockFC, :340-392 ResnetPointnet).
forward() is the reference composition (training / oracle); forward_factored() is the
inference path on the split-precision GEMM (csrc/gemm_f16x3.hip)."""
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
        from .. import gemm
        h = self.block_0.size_h
        B, T, _ = pos_term.shape
        x2 = pos_term.reshape(B * T, 2 * h)
        M = B * T
        fast = gemm.usable(M, h, 2 * h, x2) and h % 128 == 0
        if not fast:                                                         # plain torch, same factoring
            net = self.block_0(pos_term)
            for i in range(1, 5):
                blk = getattr(self, 'block_%d' % i)
                pooled = torch.relu(self.pool(net, dim=1))                   # (B,h)
                g0 = F.linear(pooled, blk.fc_0.weight[:, h:], blk.fc_0.bias)
                gs = F.linear(pooled, blk.shortcut.weight[:, h:])
                ax = torch.relu(net)
                dx = blk.fc_1(torch.relu(F.linear(ax, blk.fc_0.weight[:, :h]) + g0.unsqueeze(1)))
                net = F.linear(ax, blk.shortcut.weight[:, :h]) + gs.unsqueeze(1) + dx
            return self.fc_c(self.actvn(self.pool(net, dim=1)))

        # Split-precision GEMM path.  A block is TWO GEMMs over one row-major buffer
        # cat = [hidden | block input]:
        #   hidden = fc_0(relu(input))                       A = cat[:, h:],  C = cat[:, :h]
        #   out    = [fc_1 | shortcut](relu(cat))            A = cat (contiguous rows)
        # i.e. the shortcut rides in the K loop of the second GEMM instead of being produced by
        # the first one and re-read as a residual: -0.5 GB of traffic per block at M = 262 144 and
        # no residual epilogue.  The max-pool over a proposal's points (+ the ReLU every consumer
        # applies to it) is fused into the second GEMM's epilogue; the last block is only pooled.
        weights = self._block_weights

        fuse_pool = gemm.pool_usable(M, h, 2 * h, T)
        cat = self.__dict__.pop('_cat0', None)
        if cat is None or cat.shape[0] != M or cat.device != x2.device or cat.data_ptr() + 4 * h != x2.data_ptr():
            cat = torch.empty(M, 3 * h, device=x2.device, dtype=x2.dtype)    # caller did not use input_buffer()
            cat[:, h:] = x2
        pooled, width = None, 3 * h
        for i in range(5):
            blk, w_first, w_second, w0_pool, ws_pool = weights(i)[:5]
            g0 = gs = None
            if i:
                g0 = F.linear(pooled, w0_pool, blk.fc_0.bias)                # once per proposal
                gs = F.linear(pooled, ws_pool, blk.fc_1.bias)
            gemm.linear(cat[:, h:], w_first, bias=None if i else blk.fc_0.bias, gbias=g0, rows_per_group=T,
                        relu_in=True, out=cat[:, :h])
            last = i == 4
            nxt = None if last and fuse_pool else torch.empty(M, 2 * h, device=cat.device, dtype=cat.dtype)
            pooled = torch.zeros(B, h, device=cat.device, dtype=cat.dtype) if fuse_pool else None
            gemm.linear(cat, w_second, bias=None if i else blk.fc_1.bias, gbias=gs, rows_per_group=T,
                        relu_in=True, out=None if nxt is None else nxt[:, h:], pool=pooled,
                        store=nxt is not None)
            if pooled is None:
                pooled = torch.relu(self.pool(nxt[:, h:].reshape(B, T, h), dim=1))
            cat = nxt
        return self.fc_c(pooled)                                             # = fc_c(relu(max over the points))

    def _block_weights(self, i):
        """Block i as TWO GEMMs over one buffer [hidden | block input]: (blk, W_first, W_second, W0_pool, Ws_pool) with
        W_first = fc_0 on the per-point columns, W_second = [fc_1 | shortcut on the per-point columns], and the two
        pooled-half matrices (None for block 0, whose 2h input columns are all per-point).  Built once per parameter
        version, shared by the host threads."""
        h = self.block_0.size_h
        blk = getattr(self, 'block_%d' % i)
        key = (blk.fc_0.weight._version, blk.fc_1.weight._version, blk.shortcut.weight._version,
               blk.fc_0.weight.data_ptr())
        c = self.__dict__.setdefault('_stack_cache', {})
        if c.get(i, (None,))[0] != key:
            from .. import _lib
            with _lib.BUILD_LOCK:      # shared across host threads: built once, published before it is stored
                if c.get(i, (None,))[0] != key:
                    w0, ws = blk.fc_0.weight.detach(), blk.shortcut.weight.detach()
                    wide = i == 0                                   # block 0: all 2h input columns are per-point
                    first = (w0 if wide else w0[:, :h]).contiguous()
                    second = torch.cat([blk.fc_1.weight.detach(), ws if wide else ws[:, :h]], 1).contiguous()
                    entry = (key, first, second, None if wide else w0[:, h:].contiguous(),
                             None if wide else ws[:, h:].contiguous(),
                             # both pooled-half matrices and both biases stacked: ONE small GEMM per block
                             None if wide else torch.cat([w0[:, h:], ws[:, h:]], 0).contiguous(),
                             None if wide else torch.cat([blk.fc_0.bias.detach(), blk.fc_1.bias.detach()]).contiguous())
                    _lib.publish(second.device)
                    c[i] = entry
        return (blk,) + c[i][1:]

    def frag_usable(self, B, T):
        """can forward_frag run this shape (B proposals x T points)?"""
        from .. import gemm
        h = self.block_0.size_h
        return gemm.frag_usable(B * T, h, h, T) and h % 128 == 0

    def frag_input_buffer(self, B, T, device):
        """Block 0's [hidden | input] buffer in the frag-rows layout (gemm.frag_empty) and the channel window fc_pos
        writes (pos_embed.pos_embed_frag): -> (buffer, window)."""
        from .. import gemm
        h = self.block_0.size_h
        cat = gemm.frag_empty(B * T, 3 * h, device)
        return cat, cat[:, h // 32:]

    def forward_frag(self, cat, B, T, sa):
        """forward_factored() on fragment-ordered split activations (round 6): every activation of the encoder is
        stored ONCE as relu(x) 2^sa split into f16 (hi, lo) in the operand order of the consumer's matrix instruction
        (legal because every consumer rectifies: the in-place ReLU of layers.py:27,38-46), so no GEMM re-does the
        ReLU / scale / split of its input on the VALU (it used to happen four times per block input: two n tiles x two
        GEMMs), every load is a full 1-KiB run, and no epilogue transposes through LDS.  Same two GEMMs per block,
        same factoring of the pooled half, same fused max-pool.  `cat` = frag_input_buffer()[0] with fc_pos already
        written into its window at scale 2^sa."""
        import torch.nn.functional as F
        from .. import gemm
        h = self.block_0.size_h
        hb = h // 32
        M = B * T
        pooled = None
        pools = torch.zeros(5, B, h, device=cat.device, dtype=torch.float32)       # the five blocks' max-pool targets
        for i in range(5):
            blk, w_first, w_second, _, _, w_pool, b_pool = self._block_weights(i)
            g0 = gs = None
            if i:
                g = torch.addmm(b_pool, pooled, w_pool.t())                  # (B, 2h) once per proposal: [fc_0 | shortcut]
                g0, gs = g[:, :h], g[:, h:]                                  # column windows (row stride 2h)
            gemm.linear_frag(cat[:, hb:], w_first, bias=None if i else blk.fc_0.bias, gbias=g0, rows_per_group=T,
                             out=cat[:, :hb], sa=sa)
            last = i == 4
            nxt = None if last else gemm.frag_empty(M, 2 * h, cat.device)
            pooled = pools[i]
            gemm.linear_frag(cat, w_second, bias=None if i else blk.fc_1.bias, gbias=gs, rows_per_group=T,
                             out=None if last else nxt[:, hb:], pool=pooled, store=not last, sa=sa)
            cat = nxt
        return self.fc_c(pooled)                                             # = fc_c(relu(max over the points))

    def input_buffer(self, B, T, device):
        """(B*T, 2*hidden) view for the fc_pos output that forward_factored can use in place
        (it is the right-hand part of block 0's [hidden | input] buffer)."""
        h = self.block_0.size_h
        cat = torch.empty(B * T, 3 * h, device=device, dtype=torch.float32)
        self.__dict__['_cat0'] = cat
        return cat[:, h:]
