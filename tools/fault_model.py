"""What did the faulty MFMA of round 2's failing decoder build do?  (CPU only; profiles/r03_decoder_hazard.txt section 9)

Input: the logits of the eight launches of tools/dbg_map.py on the failing build (RFD_DBG_DUMP=dir -> outs_<pid>.npy).
A bad launch has ONE wave's 16 points of a tile off the majority value.  The decoder's arithmetic is re-stated here in
float64 on the same seeded inputs and weights (csrc/occ_decoder8.hip + occ_fold.py: residual stream H', activations split
into f16 hi (round to zero) + lo, weights into f16 hi + lo (round to nearest), three MFMA products per term), and every
single-MFMA fault is tried against the measured error of the 16 points:
    layer fc_0 / fc_1  x  block 0-4  x  output-channel tile 0-15  x  32-channel k-step 0-7  x  product hi.hi / hi.lo / lo.hi
    x  {the product missing, the product added twice}
The fault's effect on the 16 logits is propagated through the rest of the network; a candidate "explains" a bad group
when the predicted error vector matches the measured one (relative residual << 1).
    python tools/fault_model.py gpurun_out/c34/dump          (all single-MFMA faults: none matches)
    python tools/fault_model.py --fcp profiles/r03_fault_dump/outs.npz     (a dump directory / npz produced by
    RFD_DBG_DUMP=dir python tools/ab/prio_check.py; round 3's own dump is summarised in profiles/r03_fault_model.txt and
    no longer tracked -- tests/golden/F_FAULT.npz keeps one process of it)
--fcp: the hypothesis that DOES match (found through the exact zeros in some error vectors: the unaffected points are the
ones whose ReLU is off for ONE input channel of block 0).  In the tile prologue H' = row0 + Wp p, one of the three terms
of ONE channel is left out for the whole wave; every (channel, term) is tried, and for the best one the weight that WAS
used is fitted (one scalar per event)."""
import glob
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rfdnet_amd import occ_fold, synthetic  # noqa: E402
from rfdnet_amd.iscnet.occ_decoder import DecoderCBatchNorm  # noqa: E402

H = 256


def f16_rtz(x):
    """round toward zero to f16 (v_cvt_pkrtz), x >= 0 float64/32 array -> float64 values representable in f16"""
    h = x.astype(np.float16)
    up = h.astype(np.float64) > x
    h = np.where(up, np.nextafter(h, np.float16(0)), h)
    return h.astype(np.float64)


def split_act(a):
    a = a.astype(np.float32).astype(np.float64)
    hi = f16_rtz(a)
    lo = f16_rtz(a - hi)
    return hi, lo


def split_w(w):
    w = w.astype(np.float32).astype(np.float64)
    hi = w.astype(np.float16).astype(np.float64)
    lo = (w - hi).astype(np.float16).astype(np.float64)
    return hi, lo


class Model(object):
    def __init__(self):
        dec = DecoderCBatchNorm(dim=3, z_dim=32, c_dim=512, hidden_size=256)
        synthetic.load_seeded(dec, 99)
        dec = dec.eval()
        rng = np.random.default_rng(5)
        K, T = 8, 1024
        self.p = ((rng.random((K, T, 3)) - 0.5) * 1.1).astype(np.float32)
        z = torch.from_numpy(rng.normal(0, 1, (K, 32)).astype(np.float32))
        c = torch.from_numpy(rng.normal(0, 1, (K, 512)).astype(np.float32))
        sd = {k: v.detach() for k, v in dec.state_dict().items()}
        fc0, fc1 = occ_fold.stacked_fc_weights(sd)
        self.kw0 = [occ_fold.choose_kw([fc0[i]]) for i in range(5)]
        self.kw1 = occ_fold.choose_kw([fc1])
        table, fc_p_w = occ_fold.fold_table(sd, z, c, self.kw0, self.kw1, ka=occ_fold.KA)
        self.table = table.numpy().astype(np.float64)              # (K, 23, 256)
        self.fc_p_w = fc_p_w.numpy().astype(np.float64)            # (256, 3), scaled by 2^KH
        self.W0 = [split_w(np.ldexp(fc0[i].numpy().astype(np.float64), self.kw0[i])) for i in range(5)]
        self.W1 = [split_w(np.ldexp(fc1[i].numpy().astype(np.float64), self.kw1)) for i in range(5)]
        self.wo = dec.fc_out.weight.detach().reshape(-1).numpy().astype(np.float64)
        self.bo = float(dec.fc_out.bias.detach())

    # --- the network from a given point on; X has shape (..., 256) -------------------------------------------------
    def from_H(self, k, Hp, blk):
        """residual stream H' in front of block blk -> logits"""
        for i in range(blk, 5):
            acc0 = self.fc0_acc(k, Hp, i)[0]
            Hp = self.from_acc0(k, Hp, acc0, i, only_block=True)
        tb = self.table[k]
        return self.bo + np.maximum(tb[21] * Hp + tb[22], 0.0) @ self.wo

    def fc0_acc(self, k, Hp, i):
        tb = self.table[k]
        a = np.maximum(tb[1 + 4 * i] * Hp + tb[2 + 4 * i], 0.0)
        hi, lo = split_act(a)
        wh, wl = self.W0[i]
        return hi @ wh.T + lo @ wh.T + hi @ wl.T, (hi, lo)

    def from_acc0(self, k, Hp, acc0, i, only_block=False):
        tb = self.table[k]
        a2 = np.maximum(tb[3 + 4 * i] * acc0 + tb[4 + 4 * i], 0.0)
        hi, lo = split_act(a2)
        wh, wl = self.W1[i]
        Hn = Hp + hi @ wh.T + lo @ wh.T + hi @ wl.T
        if only_block:
            return Hn
        return self.from_H(k, Hn, i + 1)

    def terms(self, hi, lo, wh, wl):
        """-> array (16 tiles, 8 k-steps, 3 products, n, 16 channels): the value every MFMA adds to its accumulator"""
        n = hi.shape[0]
        out = np.zeros((16, 8, 3, n, 16))
        for ks in range(8):
            ksl = slice(32 * ks, 32 * ks + 32)
            for ti, (x, w) in enumerate(((hi, wh), (lo, wh), (hi, wl))):          # hi.hi, (W hi)(a lo), (W lo)(a hi)
                full = x[:, ksl] @ w[:, ksl].T                                      # (n, 256)
                out[:, ks, ti] = full.reshape(n, 16, 16).transpose(1, 0, 2)
        return out

    def candidates(self, k, idx):
        """-> list of (name, predicted logits (16,)) over all single-MFMA faults, + the fault-free logits"""
        P = self.p[k, idx].astype(np.float64)
        Hp = self.table[k][0] + P @ self.fc_p_w.T
        base = self.from_H(k, Hp, 0)
        names, preds = [], []
        for i in range(5):
            acc0, (hi, lo) = self.fc0_acc(k, Hp, i)
            T0 = self.terms(hi, lo, *self.W0[i])
            for sign, what in ((-1.0, "missing"), (1.0, "twice")):
                d = np.zeros((16, 8, 3) + acc0.shape)
                for t in range(16):
                    d[t, :, :, :, 16 * t:16 * t + 16] = sign * T0[t]
                lg = self.from_acc0(k, Hp, acc0 + d, i)                               # (16,8,3,n)
                for t in range(16):
                    for ks in range(8):
                        for ty in range(3):
                            names.append("fc_0 block %d out-tile %2d k-step %d %s %s" % (i, t, ks, ("hi.hi", "Whi.alo", "Wlo.ahi")[ty], what))
                            preds.append(lg[t, ks, ty])
            # fc_1 of this block
            tb = self.table[k]
            a2 = np.maximum(tb[3 + 4 * i] * acc0 + tb[4 + 4 * i], 0.0)
            h2, l2 = split_act(a2)
            T1 = self.terms(h2, l2, *self.W1[i])
            Hn = self.from_acc0(k, Hp, acc0, i, only_block=True)
            for sign, what in ((-1.0, "missing"), (1.0, "twice")):
                d = np.zeros((16, 8, 3) + Hn.shape)
                for t in range(16):
                    d[t, :, :, :, 16 * t:16 * t + 16] = sign * T1[t]
                lg = self.from_H(k, Hn + d, i + 1)
                for t in range(16):
                    for ks in range(8):
                        for ty in range(3):
                            names.append("fc_1 block %d out-tile %2d k-slab %d %s %s" % (i, t, ks, ("hi.hi", "Whi.alo", "Wlo.ahi")[ty], what))
                            preds.append(lg[t, ks, ty])
            Hp = Hn
        return base, names, np.array(preds)


def load_dumps(path):
    if path.endswith(".npz"):
        z = np.load(path)
        return [(k, z[k]) for k in sorted(z.files)]
    return [(os.path.basename(f), np.load(f)) for f in sorted(glob.glob(os.path.join(path, "outs_*.npy")))]


def bad_groups(outs):
    ref = np.median(outs, axis=0)
    for r in range(outs.shape[0]):
        bad = np.abs(outs[r] - ref) > 1e-5
        for k, g in sorted({(int(k), int(t) // 16) for k, t in np.argwhere(bad)}):
            idx = np.arange(16 * g, 16 * g + 16)
            yield r, k, g, idx, (outs[r, k, idx] - ref[k, idx]).astype(np.float64)


def fcp_scan(path):
    """Every (channel, term of fc_p) left out -> which one reproduces the measured error of the 16 points?"""
    from scipy.optimize import minimize_scalar
    m = Model()
    n = hits = 0
    tally = {}
    for name, outs in load_dumps(path):
        for r, k, g, idx, e in bad_groups(outs):
            P = m.p[k, idx].astype(np.float64)
            H0 = m.table[k][0] + P @ m.fc_p_w.T
            base = m.from_H(k, H0, 0)
            Hc = np.broadcast_to(H0, (H, 3) + H0.shape).copy()
            for j in range(3):
                for ch in range(H):
                    Hc[ch, j, :, ch] -= P[:, j] * m.fc_p_w[ch, j]
            rr = np.linalg.norm((m.from_H(k, Hc, 0) - base) - e, axis=2) / np.linalg.norm(e)
            ch, j = np.unravel_index(np.argmin(rr), rr.shape)
            second = np.partition(rr.ravel(), 1)[1]
            w = m.fc_p_w[ch, j]

            def cost(s_):
                Hx = H0.copy()
                Hx[:, ch] -= (1 - s_) * w * P[:, j]
                return np.linalg.norm((m.from_H(k, Hx, 0) - base) - e)
            o = minimize_scalar(cost, bounds=(-0.5, 0.5), method="bounded", options={"xatol": 1e-7})
            print("%s launch %d prop %d tile %d wave %d: |e| max %.1e  channel %3d (H' tile %d register %d, lanes %d-%d) term %s "
                  "left out: residual %.4f (next best %.2f); weight used / true weight = %+.1e"
                  % (name, r, k, g // 8, g % 8, np.abs(e).max(), ch, ch // 16, ch % 4, 16 * ((ch % 16) // 4), 16 * ((ch % 16) // 4) + 15,
                     "xyz"[j], rr[ch, j], second, o.x))
            n += 1
            hits += rr[ch, j] < 0.02
            tally[(int(ch), "xyz"[j])] = tally.get((int(ch), "xyz"[j]), 0) + 1
            sys.stdout.flush()
    print("# %d wrong 16-point groups, %d explained (residual < 0.02) by ONE missing fc_p term; (channel, term): count = %s" % (n, hits, tally))


def main():
    if sys.argv[1] == "--fcp":
        return fcp_scan(sys.argv[2])
    d = sys.argv[1]
    m = Model()
    shown = 0
    for f, outs in load_dumps(d):
        ref = np.median(outs, axis=0)
        for r in range(outs.shape[0]):
            bad = np.abs(outs[r] - ref) > 1e-5
            groups = sorted({(int(k), int(t) // 16) for k, t in np.argwhere(bad)})
            for k, g in groups:
                idx = np.arange(16 * g, 16 * g + 16)
                e = (outs[r, k, idx] - ref[k, idx]).astype(np.float64)
                base, names, preds = m.candidates(k, idx)
                # the re-statement itself: how far is it from the majority logits?
                model_err = np.abs(base - ref[k, idx]).max()
                res = np.linalg.norm((preds - base) - e, axis=1) / np.linalg.norm(e)
                order = np.argsort(res)[:3]
                print("%s launch %d prop %d tile %d wave %d: |e| max %.2e (model vs majority %.1e)"
                      % (f, r, k, g // 8, g % 8, np.abs(e).max(), model_err))
                for o in order:
                    print("    residual %.3f  %s" % (res[o], names[o]))
                shown += 1
                sys.stdout.flush()
    print("# %d bad groups" % shown)


if __name__ == "__main__":
    main()
