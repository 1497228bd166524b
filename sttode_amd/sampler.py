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

    @torch.no_grad()
    def forward(self, net, mean=True, need_weights=False, eps=None):
        """sampler.py:32-70 -> (dec_motion [n,K,Tf,2], sampler_dist, vae_dist, attn_weights (= net.pred_traj, sic)).
        ``eps`` ([1,nz] when share_eps else [n,nz]) may be injected; otherwise drawn like sampler.py:41-46.  (The reference
        sizes eps by net.agent_num = agents per scene, so its sampled modes only run with one scene per call; here eps is
        per agent of the whole batch, identical in that case.)"""
        if self.device.type != 'cuda':
            raise capi.SttodeError('Sampler runs only on a HIP device (no CPU fallback): call set_device(cuda) first')
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
        if mean:
            mode, eps = 0, None
        else:
            rows = 1 if self.share_eps else n
            eps = torch.randn(rows, nz, device=self.device) if eps is None else eps.to(self.device, torch.float32).contiguous()
            if tuple(eps.shape) != (rows, nz):
                raise ValueError(f'eps must be [{rows}, {nz}]')
            mode = 1 if self.share_eps else 2
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
