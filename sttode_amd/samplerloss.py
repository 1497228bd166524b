"""Stage-2 objective: drop-in for samplerloss.py:4-73 (values; the per-agent reductions run in ``sttode_sampler_loss``)."""
import torch

from . import capi


def get_diversity_config(dataset):
    """trainsampler.py:102-116."""
    return {'sdd': {'weight': 0.5, 'scale': 0.5}, 'eth': {'weight': 1, 'scale': 1}, 'univ': {'weight': 10, 'scale': 10.0},
            'nba': {'weight': 1, 'scale': 1.0}}.get(dataset, {'weight': 3, 'scale': 2})


class _PerAgentLoss(torch.autograd.Function):
    """(kld[a], div[a]) = sttode_sampler_loss; backward = sttode_sampler_loss_bwd (gradients wrt mu, logvar, motion)."""

    @staticmethod
    def forward(ctx, mu, logvar, motion, pmu, plogvar, scale):
        n, K = motion.shape[:2]
        D = motion[0, 0].numel()
        nz = mu.shape[-1]
        kld = torch.empty(n, device=mu.device)
        div = torch.empty(n, device=mu.device)
        capi.call('sttode_sampler_loss', mu, logvar, pmu, plogvar, motion, n, K, nz, D, float(scale), kld, div, capi.stream_ptr())
        ctx.save_for_backward(mu, logvar, motion, *([pmu, plogvar] if pmu is not None else []))
        ctx.dims = (n, K, nz, D, float(scale))
        return kld, div

    @staticmethod
    def backward(ctx, g_kld, g_div):
        mu, logvar, motion, *pp = ctx.saved_tensors
        pmu, plogvar = pp if pp else (None, None)
        n, K, nz, D, scale = ctx.dims
        dmu, dlv, dmo = torch.empty_like(mu), torch.empty_like(logvar), torch.empty_like(motion)
        capi.call('sttode_sampler_loss_bwd', mu, logvar, pmu, plogvar, motion, g_kld.contiguous(), g_div.contiguous(), n, K, nz, D, scale,
                  dmu, dlv, dmo, capi.stream_ptr())
        return dmu, dlv, dmo, None, None, None


def _per_agent(q, p, motion, scale):
    """(kld[a], div[a]) for motion [n,K,Tf,2], q/p Normal over [n*K, nz]; differentiable wrt q.mu, q.logvar and motion."""
    if q.mu.device.type != 'cuda':
        raise capi.SttodeError('sampler loss runs only on a HIP device (no CPU fallback)')
    f = lambda t: t.contiguous().float()
    return _PerAgentLoss.apply(f(q.mu), f(q.logvar), f(motion), None if p is None else f(p.mu).detach(),
                               None if p is None else f(p.logvar).detach(), scale)


def compute_z_kld(q_z_dist_dlow, p_z_dist_infer, agent_num, min_clip, weight, _kld=None):
    s = (_kld if _kld is not None else q_z_dist_dlow.kl(p_z_dist_infer)).sum()
    uw = (s / agent_num).clamp_min(min_clip)
    return uw * weight, uw


def diversity_loss(infer_dec_motion, agent_num, weight, scale, _div=None):
    if _div is None:
        n, K = infer_dec_motion.shape[:2]
        zeros = torch.zeros(n * K, 16, device=infer_dec_motion.device)
        from .dist import Normal
        _, _div = _per_agent(Normal(mu=zeros, logvar=zeros), None, infer_dec_motion, scale)
    uw = _div.sum() / agent_num
    return uw * weight, uw


def compute_sampler_loss(args, fut_motion_orig, infer_dec_motion, batch_size, fut_mask, p_z_dist, q_z_dist_dlow, div_cfg):
    """samplerloss.py:41-58: total = weighted clamped KL + weighted diversity (reconstruction term disabled upstream)."""
    agent_num = fut_motion_orig.shape[0]
    kld_a, div_a = _per_agent(q_z_dist_dlow, p_z_dist, infer_dec_motion, div_cfg['scale'])
    kld_loss, _ = compute_z_kld(q_z_dist_dlow, p_z_dist, agent_num, args.kld_min_clamp, args.kld_weight, _kld=kld_a)
    div_loss, _ = diversity_loss(infer_dec_motion, agent_num, div_cfg['weight'], div_cfg['scale'], _div=div_a)
    total_loss = kld_loss + div_loss
    loss_dict = {'kld': kld_loss, 'diverse': div_loss, 'recon': 0}
    return total_loss, loss_dict, dict(loss_dict)


def compute_sampler_loss_nba(args, fut_motion_orig, infer_dec_motion, batch_size, p_z_dist, q_z_dist_dlow, div_cfg):
    """samplerloss.py:60-73."""
    return compute_sampler_loss(args, fut_motion_orig, infer_dec_motion, batch_size, None, p_z_dist, q_z_dist_dlow, div_cfg)
