"""float64 restatement of DecoderCBatchNorm.forward (occ_decoder.py:110-123, layers.py:98-107, 226-242) in numpy:
the ground truth of the logit-band parity tests.  At |logit| ~ 30 two fp32 evaluations of the SAME module in different
summation orders already differ by more than 1e-4 (intermediate activations are much larger than the logit), so the
distance to the exact value -- not to one particular fp32 evaluation -- is what a kernel can be held to."""
import numpy as np


def decoder_f64(sd, p, z, c, eps=1e-5, return_amax=False):
    """sd: state_dict (numpy arrays, reference key names); p (K,T,3), z (K,Z), c (K,C) -> logits (K,T) float64"""
    g = lambda k: np.asarray(sd[k], dtype=np.float64)
    p, z, c = (np.asarray(a, dtype=np.float64) for a in (p, z, c))

    def cbn(prefix, x):                         # x (K,T,H)
        gamma = c @ g(prefix + ".conv_gamma.weight")[:, :, 0].T + g(prefix + ".conv_gamma.bias")
        beta = c @ g(prefix + ".conv_beta.weight")[:, :, 0].T + g(prefix + ".conv_beta.bias")
        # the module normalises in fp32 with sqrt(var + eps) computed in fp32; in exact arithmetic:
        nrm = (x - g(prefix + ".bn.running_mean")) / np.sqrt(g(prefix + ".bn.running_var") + eps)
        return gamma[:, None, :] * nrm + beta[:, None, :]

    amax = 0.0
    net = p @ g("fc_p.weight")[:, :, 0].T + g("fc_p.bias")
    net = net + (z @ g("fc_z.weight").T + g("fc_z.bias"))[:, None, :]
    for i in range(5):
        b = "blocks.%d." % i
        a0 = np.maximum(cbn(b + "bn_0", net), 0)
        h = a0 @ g(b + "fc_0.weight")[:, :, 0].T + g(b + "fc_0.bias")
        a1 = np.maximum(cbn(b + "bn_1", h), 0)
        dx = a1 @ g(b + "fc_1.weight")[:, :, 0].T + g(b + "fc_1.bias")
        amax = max(amax, float(a0.max()), float(a1.max()))       # what the kernel splits into f16 (hi, lo)
        net = net + dx
    a = np.maximum(cbn("bn", net), 0)
    out = a @ g("fc_out.weight")[0, :, 0] + g("fc_out.bias")[0]
    return (out, amax) if return_amax else out
