"""DecoderCBatchNorm with the reference's constructor, forward signature and
state_dict keys (models/iscnet/modules/occ_decoder.py:72-123, layers.py:51-107,
193-242), evaluated by ONE fused HIP kernel (csrc/occ_decoder.hip).

state_dict keys (verified against the reference class by
tests/golden/make_fixtures.py):  fc_p.{weight(256,3,1),bias}, fc_z.{weight
(256,Z),bias}, blocks.{0-4}.{bn_0,bn_1}.{conv_gamma,conv_beta}.{weight
(256,C,1),bias}, blocks.{i}.{bn_0,bn_1}.bn.{running_mean,running_var,
num_batches_tracked}, blocks.{i}.{fc_0,fc_1}.{weight(256,256,1),bias}, bn.*,
fc_out.{weight(1,256,1),bias}.
"""
import ctypes as C
import os

import torch
import torch.nn as nn

from .. import _lib, occ_fold

MODE_F16X3 = 3   # parity mode (fp32-class logits)
MODE_F16X1 = 1   # throughput mode
TILE = 128


class CBatchNorm1d(nn.Module):
    """Parameter container of layers.py:193-242 (conv_gamma, conv_beta, bn)."""

    def __init__(self, c_dim, f_dim, norm_method='batch_norm'):
        super().__init__()
        if norm_method != 'batch_norm':
            raise NotImplementedError("only batch_norm is on the inference path")
        self.c_dim = c_dim
        self.f_dim = f_dim
        self.norm_method = norm_method
        self.conv_gamma = nn.Conv1d(c_dim, f_dim, 1)
        self.conv_beta = nn.Conv1d(c_dim, f_dim, 1)
        self.bn = nn.BatchNorm1d(f_dim, affine=False)
        self.reset_parameters()

    def reset_parameters(self):            # layers.py:219-224
        nn.init.zeros_(self.conv_gamma.weight)
        nn.init.zeros_(self.conv_beta.weight)
        nn.init.ones_(self.conv_gamma.bias)
        nn.init.zeros_(self.conv_beta.bias)


class CResnetBlockConv1d(nn.Module):
    """Parameter container of layers.py:51-107 (size_in == size_h == size_out)."""

    def __init__(self, c_dim, size_in, size_h=None, size_out=None,
                 norm_method='batch_norm', legacy=False):
        super().__init__()
        if legacy:
            raise NotImplementedError("legacy CBN is not on the inference path")
        size_h = size_in if size_h is None else size_h
        size_out = size_in if size_out is None else size_out
        if not (size_in == size_h == size_out):
            raise NotImplementedError("shortcut blocks are not used by DecoderCBatchNorm")
        self.size_in, self.size_h, self.size_out = size_in, size_h, size_out
        self.bn_0 = CBatchNorm1d(c_dim, size_in, norm_method=norm_method)
        self.bn_1 = CBatchNorm1d(c_dim, size_h, norm_method=norm_method)
        self.fc_0 = nn.Conv1d(size_in, size_h, 1)
        self.fc_1 = nn.Conv1d(size_h, size_out, 1)
        self.shortcut = None
        nn.init.zeros_(self.fc_1.weight)   # layers.py:96


class DecoderCBatchNorm(nn.Module):
    def __init__(self, dim=3, z_dim=128, c_dim=128, hidden_size=256, n_blocks=5,
                 leaky=False, legacy=False):
        super().__init__()
        if dim != 3 or hidden_size != 256 or n_blocks != 5 or leaky or legacy:
            raise NotImplementedError(
                "the fused HIP decoder is built for dim=3, hidden_size=256, "
                "n_blocks=5, ReLU (the configuration RfD-Net instantiates, "
                "occupancy_net.py:45)")
        self.z_dim = z_dim
        if not z_dim == 0:
            self.fc_z = nn.Linear(z_dim, hidden_size)
        self.fc_p = nn.Conv1d(dim, hidden_size, 1)
        self.blocks = nn.ModuleList([CResnetBlockConv1d(c_dim, hidden_size)
                                     for _ in range(n_blocks)])
        self.bn = CBatchNorm1d(c_dim, hidden_size)
        self.fc_out = nn.Conv1d(hidden_size, 1, 1)
        self.mode = MODE_F16X3
        # which kernel: 'w4' = four 496-register waves per workgroup (csrc/occ_decoder.hip),
        # 'w8' = eight waves, two per SIMD (csrc/occ_decoder8.hip); same results
        self.kernel = os.environ.get("RFD_DECODER_KERNEL", "w8")
        # activation scale 2^ka of the split-f16 arithmetic (occ_fold.py): lowered ONCE, for good, by
        # lower_activation_scale() when a launch reports an activation beyond the f16 range at the current scale --
        # the reference's fp32 decoder (occ_decoder.py:110-123) cannot overflow, so neither may this one fail a scene
        self.ka = occ_fold.KA
        self.check_range = True        # forward() reads the stream's status word after the launch
        self._packed = None            # (packed stream, kw0, kw1, key)

    # ---- weight stream (re-packed only when the parameters change) -----------
    def _weights_key(self):
        ps = [self.blocks[i].fc_0.weight for i in range(5)] + \
             [self.blocks[i].fc_1.weight for i in range(5)]
        return tuple((p.data_ptr(), p._version, str(p.device)) for p in ps)

    def packed_weights(self):
        key = self._weights_key() + (self.kernel,)
        hit = self._packed
        if hit is not None and hit[3] == key:
            return hit[:3]
        with _lib.BUILD_LOCK:          # shared across host threads: built once, published before it is stored
            hit = self._packed
            if hit is not None and hit[3] == key:
                return hit[:3]
            sd = {k: v.detach() for k, v in self.state_dict().items()}
            fc0, fc1 = occ_fold.stacked_fc_weights(sd)
            if not fc0.is_cuda:
                raise RuntimeError("CPU not supported")
            kw0 = [occ_fold.choose_kw([fc0[i]]) for i in range(5)]
            kw1 = occ_fold.choose_kw([fc1])
            packed = torch.empty(_lib.lib().rfd_occ_packed_bytes(), dtype=torch.uint8,
                                 device=fc0.device)
            arr = (C.c_int * 5)(*kw0)
            with torch.cuda.device(fc0.device):
                pack = _lib.lib().rfd_occ_pack_weights_w8 if self.kernel == "w8" else _lib.lib().rfd_occ_pack_weights
                rc = pack(fc0.data_ptr(), fc1.data_ptr(), arr, kw1, packed.data_ptr(), _lib.current_stream())
            _lib.check(rc, "rfd_occ_pack_weights")
            _lib.publish(fc0.device)
            self._packed = (packed, kw0, kw1, key)      # one tuple, one store: readers never see a half-updated pair
        return self._packed[:3]

    def _fc_out_bias(self):
        b = self.fc_out.bias
        key = (b.data_ptr(), b._version)
        if getattr(self, "_bo_cache", (None, None))[0] != key:
            self._bo_cache = (key, float(b.detach().item()))   # one sync per weight load
        return self._bo_cache[1]

    def fold(self, z, c):
        """Per-proposal table (K,23,256) + scaled fc_p weight for codes z, c."""
        _, kw0, kw1 = self.packed_weights()
        if self.training:                  # parameters may move under us: fold from the live tensors
            sd = {k: v.detach() for k, v in self.state_dict().items()}
            if self.z_dim == 0:
                sd["fc_z.weight"] = torch.zeros(256, 0, device=c.device)
                sd["fc_z.bias"] = torch.zeros(256, device=c.device)
            return occ_fold.fold_table(sd, z, c, kw0, kw1, ka=self.ka)
        key = tuple((p.data_ptr(), p._version) for p in self.parameters()) + \
            tuple((b.data_ptr(), b._version) for b in self.buffers()) + (self.ka,)
        hit = getattr(self, "_fold_cache", None)
        if hit is None or hit[0] != key:
            with _lib.BUILD_LOCK:
                hit = getattr(self, "_fold_cache", None)
                if hit is None or hit[0] != key:
                    sd = {k: v.detach() for k, v in self.state_dict().items()}
                    consts = occ_fold.stacked_constants(sd, kw0, kw1, ka=self.ka)
                    _lib.publish(c.device)
                    hit = self._fold_cache = (key, consts)
        return occ_fold.fold_table_stacked(hit[1], z, c)

    def lower_activation_scale(self):
        """After status bit 2 (an activation * 2^ka reached the f16 limit): switch to the fallback scale, once.
        Returns True if the caller should run the launch again, False if the fallback scale is already in use
        (then the overflow is real: |activation| >= 8190)."""
        if self.ka <= occ_fold.KA_FALLBACK:
            return False
        import warnings
        warnings.warn("occupancy decoder: an activation exceeded the f16 range at scale 2^%d; re-running at 2^%d "
                      "(kept for this decoder from now on)" % (self.ka, occ_fold.KA_FALLBACK), RuntimeWarning)
        self.ka = occ_fold.KA_FALLBACK
        return True

    def can_scatter(self):
        """the eight-wave kernel can write MISE's value / known arrays itself (decode_tiles(scatter=...))"""
        return self.kernel == "w8" and getattr(self, "fuse_scatter", True)

    def decode_tiles(self, pts, tile_prop, table, fc_p_w, mode=None, tile_src=None, scatter=None):
        """pts (n_src_tiles*128,3) f32, tile_prop (n_tiles,) i32 [, tile_src
        (n_tiles,) i32: which source tile each tile reads] -> logits (n_tiles*128,).
        scatter = (lin i32 indexed like pts, values (K,n_per) f32, pstate (K,n_per) u8): the logits go
        straight to values[prop, lin] and the points are marked known (returns None)."""
        packed, _, _ = self.packed_weights()
        n_tiles = tile_prop.shape[0]
        assert pts.is_contiguous() and pts.dtype == torch.float32
        assert tile_src is not None or pts.shape[0] == n_tiles * TILE
        assert tile_prop.dtype == torch.int32 and table.is_contiguous()
        wo = self.fc_out.weight.detach().reshape(-1).contiguous()
        bo = self._fc_out_bias()
        if scatter is not None:
            lin, values, pstate = scatter
            assert self.can_scatter() and lin.dtype == torch.int32 and lin.shape[0] == pts.shape[0]
            assert values.dtype == torch.float32 and pstate.dtype == torch.uint8 and values.is_contiguous()
            with torch.cuda.device(pts.device):
                rc = _lib.lib().rfd_occ_decode_scatter_w8(
                    n_tiles, pts.data_ptr(), tile_prop.data_ptr(),
                    tile_src.data_ptr() if tile_src is not None else None,
                    packed.data_ptr(), fc_p_w.data_ptr(), table.data_ptr(), wo.data_ptr(), bo,
                    lin.data_ptr(), values.data_ptr(), pstate.data_ptr(), int(values.shape[1]),
                    self.mode if mode is None else mode, _lib.current_stream())
            _lib.check(rc, "rfd_occ_decode_scatter_w8")
            return None
        logits = torch.empty(n_tiles * TILE, dtype=torch.float32, device=pts.device)
        with torch.cuda.device(pts.device):
            decode = _lib.lib().rfd_occ_decode_w8 if self.kernel == "w8" else _lib.lib().rfd_occ_decode
            rc = decode(n_tiles, pts.data_ptr(), tile_prop.data_ptr(),
                        tile_src.data_ptr() if tile_src is not None else None,
                        packed.data_ptr(), fc_p_w.data_ptr(), table.data_ptr(),
                        wo.data_ptr(), bo, logits.data_ptr(),
                        self.mode if mode is None else mode, _lib.current_stream())
        _lib.check(rc, "rfd_occ_decode")
        return logits

    def forward(self, p, z, c, **kwargs):
        """p (B,T,3), z (B,Z), c (B,C) -> logits (B,T).  occ_decoder.py:110-123."""
        if not p.is_cuda:
            raise RuntimeError("CPU not supported")
        B, T, _ = p.shape
        if c.dim() == 3:
            c = c.squeeze(2)
        ka_used = self.ka                  # (a decoder shared by several host threads may be lowered by another one)
        table, fc_p_w = self.fold(z.float(), c.float())
        tpad = (T + TILE - 1) // TILE * TILE
        if tpad != T:
            pp = torch.zeros(B, tpad, 3, dtype=torch.float32, device=p.device)
            pp[:, :T] = p
        else:
            pp = p.contiguous().float()
        tiles_per = tpad // TILE
        tile_prop = torch.arange(B, dtype=torch.int32, device=p.device).repeat_interleave(tiles_per)
        logits = self.decode_tiles(pp.reshape(-1, 3), tile_prop, table, fc_p_w)
        if self.check_range:
            # the module's own forward is synchronous about the f16-range flag (one 4-byte read on this stream):
            # an overflow at the default scale is answered by the fallback scale, not by an exception
            with torch.cuda.device(p.device):
                st = _lib.stream_status_bits()
            if st & 2:
                with _lib.BUILD_LOCK:
                    lowered = self.lower_activation_scale()
            if st & 2 and (lowered or self.ka < ka_used):
                table, fc_p_w = self.fold(z.float(), c.float())
                logits = self.decode_tiles(pp.reshape(-1, 3), tile_prop, table, fc_p_w)
                with torch.cuda.device(p.device):
                    st = (st & ~2) | _lib.stream_status_bits()
            _lib.raise_status(st)
        return logits.view(B, tpad)[:, :T]
