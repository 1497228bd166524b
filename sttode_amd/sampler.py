"""Stage-2 latent sampler: drop-in for the reference's ``sampler.Sampler`` (sampler.py:7-73), forward values only.

Same constructor (``args``: sample_k, nz, share_eps, train_w_mean, qnet_mlp, dataset), same parameter names / shapes
(``q_mlp.affine_layers.{i}``, ``q_A``, ``q_b``, ``q_c``, ``linear``), same call ``forward(net, mean, need_weights)`` on a
``sttode_amd.STTODENet`` that has data set.  The Q-net runs on ``sttode_linear_cols`` (MFMA column chain, tanh epilogue),
the latent codes on ``sttode_sampler_latent``, both decodes on the model's HIP decoder.  No autograd graph (training the
sampler is SURVEY.md §8f rank 1 territory: backward kernels)."""
import torch
import torch.nn as nn

from . import capi
from .dist import Normal
from .ops import linear_cols


class _TanhMLP(nn.Module):  # utils/mlp.py:5-29 ('tanh' is the only activation the sampler asks for)
    def __init__(self, input_dim, hidden_dims=(128, 128)):
        super().__init__()
        self.out_dim = hidden_dims[-1]
        self.affine_layers = nn.ModuleList()
        last = input_dim
        for nh in hidden_dims:
            self.affine_layers.append(nn.Linear(last, nh))
            last = nh


class Sampler(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.device = torch.device('cpu')
        self.args = args
        self.nk, self.nz = args.sample_k, args.nz
        self.share_eps = args.share_eps
        self.train_w_mean = args.train_w_mean
        self.pred_model_dim = 64
        self.qnet_mlp = args.qnet_mlp
        if self.nz % 16 or any(h % 16 for h in self.qnet_mlp):
            raise NotImplementedError('Q-net widths must be multiples of 16 for the MFMA column kernels')
        self.q_mlp = _TanhMLP(self.pred_model_dim, self.qnet_mlp)
        self.q_A = nn.Linear(self.q_mlp.out_dim, self.nk * self.nz)
        self.q_b = nn.Linear(self.q_mlp.out_dim, self.nk * self.nz)
        self.q_c = nn.Linear(self.nk * self.nz, self.nz)
        self.linear = nn.Linear(128, 64)

    def set_device(self, device):
        self.device = torch.device(device)
        self.to(self.device)

    def forward(self, net, mean=True, need_weights=False, eps=None):
        """With autograd enabled (trainsampler.py:134-150,171-185) the outputs carry a graph to the sampler's parameters (backward
        on csrc/train.hip + csrc/sampler.hip kernels, see ``_SamplerFn``); otherwise plain values.  See ``_forward_values``."""
        if self.device.type != 'cuda':
            raise capi.SttodeError('Sampler runs only on a HIP device (no CPU fallback): call set_device(cuda) first')
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return _sampler_forward_grad(self, net, mean, eps)
        with torch.no_grad():
            return self._forward_values(net, mean, need_weights, eps)

    def _draw_eps(self, n, mean, eps):
        if mean:
            return 0, None
        rows = 1 if self.share_eps else n
        eps = torch.randn(rows, self.nz, device=self.device) if eps is None else eps.to(self.device, torch.float32).contiguous()
        if tuple(eps.shape) != (rows, self.nz):
            raise ValueError(f'eps must be [{rows}, {self.nz}]')
        return (1 if self.share_eps else 2), eps

    def _forward_values(self, net, mean=True, need_weights=False, eps=None):
        """sampler.py:32-70 -> (dec_motion [n,K,Tf,2], sampler_dist, vae_dist, attn_weights (= net.pred_traj, sic)).
        ``eps`` ([1,nz] when share_eps else [n,nz]) may be injected; otherwise drawn like sampler.py:41-46.  (The reference
        sizes eps by net.agent_num = agents per scene, so its sampled modes only run with one scene per call; here eps is
        per agent of the whole batch, identical in that case.)"""
        K, nz = self.nk, self.nz
        net.encode_history()
        if net._future is not None:
            net.fu_encoder()                                   # sampler.py:37 (sets qz_* attributes; unused below)
        pf = net.past_feature
        n = pf.shape[0]
        h = linear_cols(pf, self.linear.weight, self.linear.bias)                                   # :39
        for lin in self.q_mlp.affine_layers:
            h = linear_cols(h, lin.weight, lin.bias, act='tanh')                                    # :48
        A = linear_cols(h, self.q_A.weight, self.q_A.bias)                                          # [n, K*nz] == [n*K, nz]
        b = linear_cols(h, self.q_b.weight, self.q_b.bias)
        mode, eps = self._draw_eps(n, mean, eps)
        z = torch.empty(n * K, nz, device=self.device)
        logvar = torch.empty(n * K, nz, device=self.device)
        capi.call('sttode_sampler_latent', A, b, eps, mode, z, logvar, n, K, nz, capi.stream_ptr())   # :51,53
        z0 = linear_cols(z.view(n, K * nz), self.q_c.weight, self.q_c.bias)                         # :52
        sampler_dist = Normal(mu=b.view(n * K, nz), logvar=logvar)
        net.decoder_future_0(z0)
        net.decoder_future_1(z)                                                                     # p_z_s == z (:58-60)
        vae_dist = net.pz_dis
        dec_motion = net.diverse_pred_traj
        if self.args.dataset != 'nba':
            self.scene_orig = net.scene_orig
            dec_motion = dec_motion + net._ws['orig'][:, None, None, :]                             # per-agent scene origin (:65-66)
        return dec_motion, sampler_dist, vae_dist, net.pred_traj

    def step_annealer(self):
        pass


class _SamplerFn(torch.autograd.Function):
    """Q-net + latent codes + K-sample decode with a tape; backward returns the gradients of the sampler's parameters
    (the frozen STTODENet gets none: trainsampler.py:283 optimises ``sampler.parameters()`` only)."""

    @staticmethod
    def forward(ctx, smp, net, mode, eps, *params):
        from .training import Engine
        eng = getattr(net, '_engine', None)
        if eng is None or eng.dev != net.device:
            eng = net._engine = Engine(net)
        eng.multi, eng._hold = False, []                            # single chain: no side streams
        eng._enter(-1)
        eng.P = {k: v for k, v in net.named_parameters()}
        K, nz, a = smp.nk, smp.nz, net.args
        pf = net.past_feature
        n = pf.shape[0]
        h0 = eng.lin(pf, smp.linear.weight, smp.linear.bias)                                        # sampler.py:39
        hs = [h0]
        for lin in smp.q_mlp.affine_layers:
            hs.append(eng.lin(hs[-1], lin.weight, lin.bias, act='tanh'))                            # :48
        A = eng.lin(hs[-1], smp.q_A.weight, smp.q_A.bias)
        b = eng.lin(hs[-1], smp.q_b.weight, smp.q_b.bias)
        z, logvar = eng.new(n * K, nz), eng.new(n * K, nz)
        capi.call('sttode_sampler_latent', A, b, eps, mode, z, logvar, n, K, nz, eng.st)
        z0 = eng.lin(z.view(n, K * nz), smp.q_c.weight, smp.q_c.bias)                               # :52 (feeds pred_traj only)
        net.decoder_future_0(z0)
        Tp = a.past_length
        past = net._ws['xpad'][:, :2 * Tp].reshape(n, Tp, 2).contiguous()
        d = eng.decoder_fwd(pf, z, K, past, net._ws['cur'], False)
        ctx.eng, ctx.smp, ctx.d, ctx.hs, ctx.A, ctx.eps, ctx.mode, ctx.pf = eng, smp, d, hs, A, eps, mode, pf
        dec = d['pred'].view(n, K, a.future_length, 2)
        net.diverse_pred_traj = dec
        if a.dataset != 'nba':
            dec = dec + net._ws['orig'][:, None, None, :]                                           # :63-67
        return dec.contiguous(), b.view(n * K, nz), logvar

    @staticmethod
    def backward(ctx, g_dec, g_mu, g_lv):
        from .training import EW_AXPY, EW_LATENT_BWD, EW_TANH_BWD
        eng, smp, d, hs, A = ctx.eng, ctx.smp, ctx.d, ctx.hs, ctx.A
        n, K, nz = hs[0].shape[0], smp.nk, smp.nz
        m = n * K
        eng.multi = False
        eng._enter(-1)
        eng._grad_views()
        eng.param_grads = False                                     # the STTODENet is frozen in stage 2
        try:
            dz = eng.zeros(m, nz)
            if g_dec is not None:
                eng.decoder_bwd(d, g_dec.contiguous().view(m, -1), None, eng.zeros(n, 128), dz)
        finally:
            eng.param_grads = True
        g_lv = eng.zeros(m, nz) if g_lv is None else g_lv.contiguous()
        dA = eng.new(n, K * nz)
        eng.ew(EW_LATENT_BWD, dz, g_lv, A, ctx.eps, dA, i0=nz * 4 + ctx.mode, f0=K * nz)
        db = dz.view(n, K * nz)
        if g_mu is not None:
            eng.ew(EW_AXPY, db, g_mu.contiguous(), f0=1.0)
        G = {k: torch.zeros_like(p) for k, p in smp.named_parameters() if not k.startswith('q_c.')}
        dh = eng.lin_bwd(dA, smp.q_A.weight, hs[-1], G['q_A.weight'], G['q_A.bias'])
        eng.lin_bwd(db, smp.q_b.weight, hs[-1], G['q_b.weight'], G['q_b.bias'], out=dh, accumulate=True)
        for i in range(len(smp.q_mlp.affine_layers) - 1, -1, -1):
            lin = smp.q_mlp.affine_layers[i]
            eng.ew(EW_TANH_BWD, dh, dh, hs[i + 1])
            dh = eng.lin_bwd(dh, lin.weight, hs[i], G[f'q_mlp.affine_layers.{i}.weight'], G[f'q_mlp.affine_layers.{i}.bias'])
        eng.wgrad(dh, ctx.pf, G['linear.weight'], G['linear.bias'])
        return (None, None, None, None) + tuple(G.get(k) for k, _ in smp.named_parameters())


def _sampler_forward_grad(smp, net, mean, eps):
    with torch.no_grad():
        net.encode_history()
        if net._future is not None:
            net.fu_encoder()
    mode, eps = smp._draw_eps(net.past_feature.shape[0], mean, eps)
    dec, mu, logvar = _SamplerFn.apply(smp, net, mode, eps, *[p for _, p in smp.named_parameters()])
    if smp.args.dataset != 'nba':
        smp.scene_orig = net.scene_orig
    return dec, Normal(mu=mu, logvar=logvar), net.pz_dis, net.pred_traj
