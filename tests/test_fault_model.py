"""CPU: tools/fault_model.py against the committed dump of round 2's failing decoder build
(profiles/r03_fault_dump/outs.npz: logits of 8 processes x 8 launches on an MI355X).
Two things are pinned: (1) the float64 re-statement of the decoder's folded arithmetic (occ_fold.py + the kernel's
f16 hi/lo splits) reproduces the GPU's majority logits -- GPU-produced golden values for the host-side fold; (2) every
wrong 16-point group checked is explained by ONE missing y term of fc_p in one of three channels
(profiles/r03_decoder_hazard.txt section 8)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
DUMP = os.path.join(ROOT, "profiles", "r03_fault_dump", "outs.npz")


def test_restatement_matches_the_gpu_and_the_fault_is_one_missing_fc_p_term():
    import fault_model as fm
    m = fm.Model()
    name, outs = fm.load_dumps(DUMP)[0]
    ref = np.median(outs, axis=0)
    # (1) golden: 4 x 64 points of different proposals / tiles
    for k, t0 in ((0, 0), (3, 320), (5, 640), (7, 960)):
        idx = np.arange(t0, t0 + 64)
        P = m.p[k, idx].astype(np.float64)
        lg = m.from_H(k, m.table[k][0] + P @ m.fc_p_w.T, 0)
        assert np.abs(lg - ref[k, idx]).max() < 5e-6
    # (2) the first three wrong groups of the first process
    n = 0
    for r, k, g, idx, e in fm.bad_groups(outs):
        P = m.p[k, idx].astype(np.float64)
        H0 = m.table[k][0] + P @ m.fc_p_w.T
        base = m.from_H(k, H0, 0)
        best = (1e9, None)
        for ch in range(224, 256):               # H' tiles 14 and 15 (the full scan over 256 x 3 is the tool's job)
            for j in range(3):
                Hx = H0.copy()
                Hx[:, ch] -= P[:, j] * m.fc_p_w[ch, j]
                res = np.linalg.norm((m.from_H(k, Hx, 0) - base) - e) / np.linalg.norm(e)
                best = min(best, (res, (ch, j)))
        assert best[0] < 0.02 and best[1][0] in (236, 252, 254) and best[1][1] == 1, best
        n += 1
        if n == 3:
            break
    assert n == 3
