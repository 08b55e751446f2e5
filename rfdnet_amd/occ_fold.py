"""Host-side folding for the fused occupancy decoder (pure torch, device
agnostic; the arithmetic the HIP kernel expects is defined HERE).

Reference semantics being folded (all eval mode):
  CBatchNorm1d      layers.py:226-242   out = gamma(c) * BN(x) + beta(c)
                    gamma = conv_gamma(c), beta = conv_beta(c) (Conv1d k=1),
                    BN(x) = (x - running_mean) / sqrt(running_var + 1e-5)
  CResnetBlockConv1d layers.py:98-107   net = fc_0(relu(bn_0(x,c)));
                    dx = fc_1(relu(bn_1(net,c)));  x + dx
  DecoderCBatchNorm occ_decoder.py:110-123

Per proposal and CBN layer this is an affine map per channel:
  scale = gamma / sqrt(var + eps),  shift = beta - mean * scale.
The kernel keeps the residual stream as H' = (h - cumb_i) * 2^KH (cumb_i = sum
of the fc_1 biases of blocks < i), activations as a' = a * 2^ka and weights as
w' = w * 2^kw (powers of two: exact), so with acc = sum w' a':

  row 0          (fc_p.bias + fc_z(z)) * 2^KH
  S0'_i = scale0 * 2^(ka-KH)       T0'_i = (shift0 + scale0*cumb_i) * 2^ka
  S1'_i = scale1 * 2^(-kw0_i)      T1'_i = (shift1 + scale1*fc_0.bias) * 2^ka
  Sf'   = scalef * 2^(-KH)         Tf'   = shiftf + scalef*cumb_5
with KH = ka + kw1, so fc_1's MFMA accumulates straight into H'.
"""
import math

import torch

KA = 6               # default activation scale 2^6: f16 lo parts stay normal down to |a| ~ 2^-9
KA_FALLBACK = 3      # scale after an f16-range flag (status bit 2): activations up to 8190 instead of 1023
BN_EPS = 1e-5
N_BLOCKS = 5
HIDDEN = 256
TABLE_ROWS = 23


def choose_kw(weights):
    """Largest power-of-two scale keeping max|w| * 2^kw <= 2^14 (f16 headroom
    for hi, normal-range lo for all but negligible weights)."""
    mx = max(float(w.abs().max()) for w in weights)
    if not math.isfinite(mx) or mx <= 0.0:
        return 0
    kw = int(math.floor(math.log2(16384.0 / mx)))
    return max(-8, min(kw, 24))


def cbn_scale_shift(c, conv_gamma_w, conv_gamma_b, conv_beta_w, conv_beta_b,
                    mean, var):
    """c (K,C) -> scale, shift (K,H).  Conv1d(k=1) weights are (H,C,1)."""
    gw = conv_gamma_w.reshape(conv_gamma_w.shape[0], -1)
    bw = conv_beta_w.reshape(conv_beta_w.shape[0], -1)
    gamma = torch.addmm(conv_gamma_b, c, gw.t())
    beta = torch.addmm(conv_beta_b, c, bw.t())
    scale = gamma / torch.sqrt(var + BN_EPS)
    shift = beta - mean * scale
    return scale, shift


def fold_table(sd, z, c, kw0, kw1, prefix="", ka=KA):
    """Build the (K,23,256) fp32 table and the scaled fc_p weight.

    sd: state_dict-like mapping with the reference's DecoderCBatchNorm key
    names; z (K,Z), c (K,C) fp32 tensors on the same device as sd's tensors.
    Returns (table, fc_p_w_scaled (256,3))."""
    def g(k):
        return sd[prefix + k]

    K = c.shape[0]
    KA = ka
    KH = KA + kw1
    zb = torch.addmm(g("fc_z.bias"), z, g("fc_z.weight").t()) if z.shape[1] > 0 \
        else torch.zeros(K, HIDDEN, device=c.device, dtype=c.dtype)
    rows = [(g("fc_p.bias")[None, :] + zb) * (2.0 ** KH)]
    cumb = torch.zeros(HIDDEN, device=c.device, dtype=c.dtype)

    def cbn(name):
        return cbn_scale_shift(c, g(name + ".conv_gamma.weight"), g(name + ".conv_gamma.bias"),
                               g(name + ".conv_beta.weight"), g(name + ".conv_beta.bias"),
                               g(name + ".bn.running_mean"), g(name + ".bn.running_var"))

    for i in range(N_BLOCKS):
        s0, t0 = cbn("blocks.%d.bn_0" % i)
        s1, t1 = cbn("blocks.%d.bn_1" % i)
        b0 = g("blocks.%d.fc_0.bias" % i)
        rows.append(s0 * (2.0 ** (KA - KH)))
        rows.append((t0 + s0 * cumb[None, :]) * (2.0 ** KA))
        rows.append(s1 * (2.0 ** (-kw0[i])))
        rows.append((t1 + s1 * b0[None, :]) * (2.0 ** KA))
        cumb = cumb + g("blocks.%d.fc_1.bias" % i)
    sf, tf = cbn("bn")
    rows.append(sf * (2.0 ** (-KH)))
    rows.append(tf + sf * cumb[None, :])
    table = torch.stack(rows, dim=1).contiguous()
    assert table.shape == (K, TABLE_ROWS, HIDDEN)
    fc_p_w = (g("fc_p.weight").reshape(HIDDEN, 3) * (2.0 ** KH)).contiguous()
    return table, fc_p_w


def stacked_constants(sd, kw0, kw1, prefix="", ka=KA):
    """Everything of fold_table() that does not depend on the codes, stacked over
    the 11 CBN layers (order: blocks.i.bn_0, blocks.i.bn_1 for i<5, then bn) so
    that the per-scene fold is ONE (K,C)x(C,22*256) GEMM and a handful of
    elementwise kernels instead of 22 single-tile GEMMs and ~150 tiny launches."""
    def g(k):
        return sd[prefix + k]

    names = []
    for i in range(N_BLOCKS):
        names += ["blocks.%d.bn_0" % i, "blocks.%d.bn_1" % i]
    names.append("bn")
    KA = ka
    KH = KA + kw1
    wg = [g(n + ".conv_gamma.weight").reshape(HIDDEN, -1) for n in names]
    wb = [g(n + ".conv_beta.weight").reshape(HIDDEN, -1) for n in names]
    bg = [g(n + ".conv_gamma.bias") for n in names]
    bb = [g(n + ".conv_beta.bias") for n in names]
    dev, dt = wg[0].device, wg[0].dtype
    extra, smul, tmul = [], [], []
    cumb = torch.zeros(HIDDEN, device=dev, dtype=dt)
    for i in range(N_BLOCKS):
        extra += [cumb, g("blocks.%d.fc_0.bias" % i)]
        smul += [2.0 ** (KA - KH), 2.0 ** (-kw0[i])]
        tmul += [2.0 ** KA, 2.0 ** KA]
        cumb = cumb + g("blocks.%d.fc_1.bias" % i)
    extra.append(cumb)
    smul.append(2.0 ** (-KH))
    tmul.append(1.0)
    return {
        "w": torch.cat(wg + wb, dim=0).contiguous(),               # (22*256, C)
        "b": torch.cat(bg + bb, dim=0).contiguous(),
        "sqrtv": torch.stack([torch.sqrt(g(n + ".bn.running_var") + BN_EPS) for n in names]),
        "mean": torch.stack([g(n + ".bn.running_mean") for n in names]),
        "extra": torch.stack(extra),
        "smul": torch.tensor(smul, device=dev, dtype=dt).view(1, -1, 1),
        "tmul": torch.tensor(tmul, device=dev, dtype=dt).view(1, -1, 1),
        "fc_p_b": g("fc_p.bias"),
        "fc_p_w": (g("fc_p.weight").reshape(HIDDEN, 3) * (2.0 ** KH)).contiguous(),
        "fc_z_w": g("fc_z.weight") if (prefix + "fc_z.weight") in sd else None,
        "fc_z_b": g("fc_z.bias") if (prefix + "fc_z.bias") in sd else None,
        "row0_mul": 2.0 ** KH,
    }


def fold_table_stacked(k, z, c):
    """fold_table() on the stacked constants: same per-element arithmetic (only the
    GEMM summation order differs).  Returns (table (K,23,256), fc_p_w_scaled)."""
    K = c.shape[0]
    L = k["mean"].shape[0]
    gb = torch.addmm(k["b"], c, k["w"].t()).view(K, 2 * L, HIDDEN)
    if c.is_cuda and c.dtype == torch.float32 and not torch.is_grad_enabled():
        # the elementwise part as ONE kernel (csrc/small_ops.hip rfd_occ_fold_rows): same operations, same order, every
        # one rounded on its own -- the table is bit-identical to the torch composition below (nine launches)
        from . import _lib
        if k["fc_z_w"] is not None and z.shape[1] > 0:
            row0 = (k["fc_p_b"][None, :] + torch.addmm(k["fc_z_b"], z, k["fc_z_w"].t())) * k["row0_mul"]
            stride = HIDDEN
        else:
            row0 = (k["fc_p_b"] * k["row0_mul"])[None, :].contiguous()
            stride = 0
        table = torch.empty(K, TABLE_ROWS, HIDDEN, device=c.device, dtype=c.dtype)
        smul, tmul = k["smul"].reshape(-1), k["tmul"].reshape(-1)
        with torch.cuda.device(c.device):
            rc = _lib.lib().rfd_occ_fold_rows(K, L, HIDDEN, gb.data_ptr(), k["sqrtv"].data_ptr(), k["mean"].data_ptr(),
                                              k["extra"].data_ptr(), smul.data_ptr(), tmul.data_ptr(), row0.data_ptr(),
                                              stride, table.data_ptr(), _lib.current_stream())
        _lib.check(rc, "rfd_occ_fold_rows")
        return table, k["fc_p_w"]
    scale = gb[:, :L] / k["sqrtv"]
    shift = gb[:, L:] - k["mean"] * scale
    table = torch.empty(K, TABLE_ROWS, HIDDEN, device=c.device, dtype=c.dtype)
    if k["fc_z_w"] is not None and z.shape[1] > 0:
        table[:, 0] = (k["fc_p_b"][None, :] + torch.addmm(k["fc_z_b"], z, k["fc_z_w"].t())) * k["row0_mul"]
    else:
        table[:, 0] = (k["fc_p_b"] * k["row0_mul"])[None, :]
    table[:, 1::2] = scale * k["smul"]
    table[:, 2::2] = (shift + scale * k["extra"]) * k["tmul"]
    return table, k["fc_p_w"]


def stacked_fc_weights(sd, prefix=""):
    """(5,256,256) fc_0 and fc_1 weight stacks (Conv1d kernels squeezed)."""
    fc0 = torch.stack([sd[prefix + "blocks.%d.fc_0.weight" % i].reshape(HIDDEN, HIDDEN)
                       for i in range(N_BLOCKS)]).contiguous()
    fc1 = torch.stack([sd[prefix + "blocks.%d.fc_1.weight" % i].reshape(HIDDEN, HIDDEN)
                       for i in range(N_BLOCKS)]).contiguous()
    return fc0, fc1
