"""Diagonal Gaussian used by the latent model and the stage-2 sampler.

API-compatible with the reference's ``Normal`` (utils/dist.py:5-36 / model/STTODE.py:79-109): construct from ``mu`` and
``logvar`` or from a packed ``params`` tensor (mu | logvar along the last axis); attributes ``mu``, ``logvar``, ``sigma``;
``rsample`` / ``sample`` / ``mode`` / ``kl``.  These are element-wise conveniences on device tensors -- the reductions that
matter for the objectives run in ``sttode_loss_kl`` / ``sttode_sampler_loss`` (csrc/train.hip, csrc/sampler.hip).
"""
import torch

_EPS = 1e-8   # added to the prior's sigma in the two-distribution KL (utils/dist.py:26-27)


def _split_params(params):
    half = params.shape[-1] // 2
    return params[..., :half], params[..., half:]


class Normal:
    __slots__ = ('mu', 'logvar', 'sigma')

    def __init__(self, mu=None, logvar=None, params=None):
        if params is not None:
            mu, logvar = _split_params(params)
        if mu is None or logvar is None:
            raise ValueError('Normal needs (mu, logvar) or params')
        self.mu, self.logvar = mu, logvar
        self.sigma = (logvar * 0.5).exp()

    # sampling ---------------------------------------------------------------------------------
    def rsample(self):
        noise = torch.randn_like(self.sigma)
        return torch.addcmul(self.mu, noise, self.sigma)

    sample = rsample

    def mode(self):
        return self.mu

    # divergence -------------------------------------------------------------------------------
    def kl(self, p=None):
        """Element-wise KL(self || p); ``p=None`` means the standard normal in closed form."""
        if p is None:
            return 0.5 * (self.mu.square() + self.logvar.exp() - self.logvar - 1.0)
        prior_sigma = p.sigma + _EPS
        shift = (self.mu - p.mu) / prior_sigma
        ratio = self.sigma / prior_sigma
        return 0.5 * (shift.square() + ratio.square()) - 0.5 - ratio.log()
