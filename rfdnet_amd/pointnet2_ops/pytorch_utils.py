"""`pointnet2_ops.pytorch_utils` (reference external/pointnet2_ops_lib/pointnet2_ops/pytorch_utils.py:6-45):
the BatchNorm-momentum schedule `models/optimizers.py:5` imports.  Training-side only (the trainer is out of
scope); it exists so that every import the reference makes of the package resolves here."""
import torch.nn as nn

_BN_TYPES = (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d)


def set_bn_momentum_default(bn_momentum):
    """-> a function for `Module.apply` that writes `bn_momentum` into every BatchNorm layer."""
    def visit(module):
        if isinstance(module, _BN_TYPES):
            module.momentum = bn_momentum
    return visit


class BNMomentumScheduler(object):
    """momentum(epoch) = bn_lambda(epoch), applied to every BatchNorm of `model` on each `step`.
    The constructor applies epoch `last_epoch + 1` and then records `last_epoch` unchanged, as the
    reference does (pytorch_utils.py:33-34)."""

    def __init__(self, cfg, model, bn_lambda, last_epoch=-1, setter=set_bn_momentum_default):
        if not isinstance(model, nn.Module):
            raise RuntimeError("Class '{}' is not a PyTorch nn Module".format(type(model).__name__))
        self.cfg, self.model, self.setter, self.lmbd = cfg, model, setter, bn_lambda
        self.step(last_epoch + 1)
        self.last_epoch = last_epoch

    def step(self, epoch=None):
        self.last_epoch = self.last_epoch + 1 if epoch is None else epoch
        self.model.apply(self.setter(self.lmbd(self.last_epoch)))

    def show_momentum(self):
        self.cfg.log_string('Current BN decay momentum :%f.' % (self.lmbd(self.last_epoch)))
