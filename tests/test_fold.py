"""CPU: the host-side folding (rfdnet_amd/occ_fold.py) + a numpy emulation of
the arithmetic the HIP decoder performs (f16 hi/lo splits, power-of-two
scalings, residual stream kept as H') must reproduce the oracle decoder."""
import os
from collections import OrderedDict

import numpy as np
import torch

from rfdnet_amd import occ_fold, synthetic


def rtz_f16(x):
    """float32 -> float16 round-toward-zero (v_cvt_pkrtz_f16_f32)."""
    h = x.astype(np.float16)
    over = np.abs(h.astype(np.float32)) > np.abs(x)
    h[over] = np.nextafter(h[over], np.float16(0))
    return h


def split_act(a):
    hi = rtz_f16(a)
    lo = rtz_f16(a - hi.astype(np.float32))
    return hi.astype(np.float64), lo.astype(np.float64)


def split_w(w):
    hi = w.astype(np.float16)
    lo = (w - hi.astype(np.float32)).astype(np.float16)
    return hi.astype(np.float64), lo.astype(np.float64)


def emulate(table, fc_p_w, fc0, fc1, kw0, kw1, wout, bout, pts, terms=3):
    """table (23,256) of ONE proposal; pts (T,3).  fp64 accumulation stands in
    for the MFMA's fp32 accumulate (the split error dominates)."""
    H = (pts.astype(np.float32) @ fc_p_w.T.astype(np.float32) + table[0]).astype(np.float32)
    for i in range(5):
        S0, T0, S1, T1 = table[1 + 4 * i: 5 + 4 * i]
        a = np.maximum(S0 * H + T0, 0).astype(np.float32)
        w0h, w0l = split_w(np.ldexp(fc0[i], kw0[i]).astype(np.float32))
        ah, al = split_act(a)
        acc = ah @ w0h.T
        if terms == 3:
            acc = acc + ah @ w0l.T + al @ w0h.T
        a2 = np.maximum(S1 * acc.astype(np.float32) + T1, 0).astype(np.float32)
        w1h, w1l = split_w(np.ldexp(fc1[i], kw1).astype(np.float32))
        bh, bl = split_act(a2)
        upd = bh @ w1h.T
        if terms == 3:
            upd = upd + bh @ w1l.T + bl @ w1h.T
        H = (H.astype(np.float64) + upd).astype(np.float32)
    Sf, Tf = table[21], table[22]
    a = np.maximum(Sf * H + Tf, 0).astype(np.float32)
    return (a.astype(np.float64) @ wout.astype(np.float64) + bout).astype(np.float32)


def _setup(golden_dir):
    fx = np.load(os.path.join(golden_dir, "F_DEC.npz"))
    shapes = OrderedDict((str(n), tuple(int(x) for x in str(s).split(",")) if str(s) else ())
                         for n, s in zip(fx["names"], fx["shapes"]))
    sd_np = synthetic.seeded_state_dict(shapes, int(fx["seed"]))
    sd = {k: torch.from_numpy(v) for k, v in sd_np.items()}
    return fx, sd


def test_fold_plus_split_emulation_matches_reference_logits(golden_dir):
    fx, sd = _setup(golden_dir)
    fc0, fc1 = occ_fold.stacked_fc_weights(sd)
    kw0 = [occ_fold.choose_kw([fc0[i]]) for i in range(5)]
    kw1 = occ_fold.choose_kw([fc1])
    table, fc_p_w = occ_fold.fold_table(sd, torch.from_numpy(fx["z"]), torch.from_numpy(fx["c"]), kw0, kw1)
    assert table.shape == (3, 23, 256)
    wout = sd["fc_out.weight"].reshape(-1).numpy()
    bout = float(sd["fc_out.bias"])
    for k in range(3):
        out = emulate(table[k].numpy(), fc_p_w.numpy(), fc0.numpy(), fc1.numpy(), kw0, kw1,
                      wout, bout, fx["p"][k][:256])
        err = np.abs(out - fx["logits"][k][:256]).max()
        assert err < 2e-5, (k, err)


def test_single_term_mode_is_coarser_but_sane(golden_dir):
    fx, sd = _setup(golden_dir)
    fc0, fc1 = occ_fold.stacked_fc_weights(sd)
    kw0 = [occ_fold.choose_kw([fc0[i]]) for i in range(5)]
    kw1 = occ_fold.choose_kw([fc1])
    table, fc_p_w = occ_fold.fold_table(sd, torch.from_numpy(fx["z"]), torch.from_numpy(fx["c"]), kw0, kw1)
    out = emulate(table[0].numpy(), fc_p_w.numpy(), fc0.numpy(), fc1.numpy(), kw0, kw1,
                  sd["fc_out.weight"].reshape(-1).numpy(), float(sd["fc_out.bias"]), fx["p"][0][:256], terms=1)
    err = np.abs(out - fx["logits"][0][:256]).max()
    assert 1e-6 < err < 5e-3, err


def test_choose_kw_keeps_f16_headroom():
    w = torch.tensor([[0.0625, -0.03], [1e-4, 0.01]])
    kw = occ_fold.choose_kw([w])
    assert float(w.abs().max()) * 2.0 ** kw <= 16384.0 < float(w.abs().max()) * 2.0 ** (kw + 1)
    assert occ_fold.choose_kw([torch.zeros(2, 2)]) == 0


def test_stacked_fold_equals_layerwise_fold(golden_dir):
    """The per-scene fast path (one GEMM over all 11 CBN layers) produces the same
    table as the layer-by-layer definition."""
    fx, sd = _setup(golden_dir)
    fc0, fc1 = occ_fold.stacked_fc_weights(sd)
    kw0 = [occ_fold.choose_kw([fc0[i]]) for i in range(5)]
    kw1 = occ_fold.choose_kw([fc1])
    z, c = torch.from_numpy(fx["z"]), torch.from_numpy(fx["c"])
    t_ref, w_ref = occ_fold.fold_table(sd, z, c, kw0, kw1)
    t, w = occ_fold.fold_table_stacked(occ_fold.stacked_constants(sd, kw0, kw1), z, c)
    assert torch.equal(w, w_ref)
    # identical arithmetic per element; only the GEMM summation order may differ
    assert torch.allclose(t, t_ref, rtol=1e-5, atol=1e-5 * float(t_ref.abs().max()))


def test_folded_layers_are_built_once_when_several_host_threads_share_a_network():
    """bench.py runs several scenes in flight on ONE model: the lazily built caches (here: a folded conv + BatchNorm)
    are built under rfdnet_amd._lib.BUILD_LOCK -- eight threads asking at the same moment all get the same tensors."""
    import threading
    import torch
    from rfdnet_amd import fold_bn
    conv = torch.nn.Conv1d(16, 32, 1)
    bn = torch.nn.BatchNorm1d(32).eval()
    with torch.no_grad():
        bn.running_mean.normal_()
        bn.running_var.uniform_(0.5, 1.5)
    out, go = [None] * 8, threading.Barrier(8)

    def work(i):
        go.wait()
        out[i] = fold_bn.folded(conv, bn)
    ts = [threading.Thread(target=work, args=(i,)) for i in range(8)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert all(o[0] is out[0][0] and o[1] is out[0][1] for o in out)
    ref = conv.weight[:, :, 0] * (bn.weight / torch.sqrt(bn.running_var + bn.eps))[:, None]
    assert torch.allclose(out[0][0], ref.detach(), atol=1e-6)
