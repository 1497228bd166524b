"""Diagonal normal used by the latent model and the stage-2 sampler (mirror of utils/dist.py:5-36 / model/STTODE.py:79-109).

Element-wise bookkeeping on device tensors; the reductions that matter for the sampler objective run in
``sttode_sampler_loss`` (csrc/sampler.hip)."""
import torch


class Normal:
    def __init__(self, mu=None, logvar=None, params=None):
        if params is not None:
            self.mu, self.logvar = torch.chunk(params, chunks=2, dim=-1)
        else:
            assert mu is not None
            assert logvar is not None
            self.mu, self.logvar = mu, logvar
        self.sigma = torch.exp(0.5 * self.logvar)

    def rsample(self):
        return self.mu + torch.randn_like(self.sigma) * self.sigma

    def sample(self):
        return self.rsample()

    def kl(self, p=None):
        """KL(q || p), element-wise (utils/dist.py:22-30)."""
        if p is None:
            return -0.5 * (1 + self.logvar - self.mu.pow(2) - self.logvar.exp())
        t1 = (self.mu - p.mu) / (p.sigma + 1e-8)
        t2 = self.sigma / (p.sigma + 1e-8)
        return 0.5 * (t1 * t1 + t2 * t2) - 0.5 - torch.log(t2)

    def mode(self):
        return self.mu
