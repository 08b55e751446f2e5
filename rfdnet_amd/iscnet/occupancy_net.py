"""Occupancy network wrapper (models/iscnet/modules/occupancy_net.py:12-189,
generation path only): holds the decoder under the reference's attribute name
(`decoder` => state_dict keys completion.decoder.*), the prior over z and the
mesh generator."""
import torch
import torch.distributions as dist
import torch.nn as nn

from .generator import Generator3D
from .occ_decoder import DecoderCBatchNorm
from .registers import MODULES


@MODULES.register_module
class ONet(nn.Module):
    def __init__(self, cfg, optim_spec=None):
        super().__init__()
        self.optim_spec = optim_spec
        data = cfg.config['data']
        self.z_dim = data['z_dim']
        self.use_cls_for_completion = data['use_cls_for_completion']
        base = data['c_dim'] if data['skip_propagate'] else 128
        c_dim = self.use_cls_for_completion * cfg.dataset_config.num_class + base
        self.threshold = data['threshold']
        # the latent encoder q(z|p,occ,c) (encoder_latent.py) is training-only:
        # generation uses the prior mean (occupancy_net.py:138-143)
        self.encoder_latent = None
        self.decoder = DecoderCBatchNorm(dim=3, z_dim=self.z_dim, c_dim=c_dim)
        gen = cfg.config.get('generation')
        if gen and gen['generate_mesh']:
            self.generator = Generator3D(
                self, threshold=data['threshold'], resolution0=gen['resolution_0'],
                upsampling_steps=gen['upsampling_steps'], sample=gen['use_sampling'],
                refinement_step=gen['refinement_step'], simplify_nfaces=gen['simplify_nfaces'],
                preprocessor=None)

    def get_prior_z(self, z_dim, device):
        return dist.Normal(torch.zeros(z_dim, device=device), torch.ones(z_dim, device=device))

    def get_z_from_prior(self, size=torch.Size([]), device='cuda', sample=False):
        p0_z = self.get_prior_z(self.z_dim, device)
        if sample:
            return p0_z.sample(size)
        z = p0_z.mean
        return z.expand(*size, *z.size())

    def decode(self, input_points_for_completion, z, features, **kwargs):
        """-> Bernoulli over occupancy, logits (B,T)  (occupancy_net.py:147-156)"""
        logits = self.decoder(input_points_for_completion, z, features, **kwargs)
        return dist.Bernoulli(logits=logits)

    def forward(self, input_points_for_completion, input_features_for_completion,
                cls_codes_for_completion, sample=False, **kwargs):
        device = input_features_for_completion.device
        if self.use_cls_for_completion:
            input_features_for_completion = torch.cat(
                [input_features_for_completion, cls_codes_for_completion.to(device).float()], dim=-1)
        z = self.get_z_from_prior((input_points_for_completion.size(0),), device, sample=sample)
        return self.decode(input_points_for_completion, z, input_features_for_completion, **kwargs)
