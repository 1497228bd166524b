"""CPU oracle: PyTorch-eager fp32 restatement of the STTODE forward hot path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Parity status: PINNED —
tests/test_oracle_golden.py checks every function here against vectors produced
by importing the reference itself (tests/golden/make_golden.py) with weights from
sttode_amd.weights.make_weights.

The module tree reproduces the reference's ``state_dict`` surface exactly
(SURVEY.md §8b), so one weight set loads into the reference, into this oracle
and into the HIP product module.  All file:line citations are into
/root/reference.

Differences from the reference that are deliberate (and output-neutral):
  * device agnostic (no ``.cuda()``; model/STTODE.py:333-334, hypertransformer.py:69);
  * the latent ``z`` can be injected (reference draws it with torch.randn_like,
    model/STTODE.py:92) so that parity tests are deterministic;
  * torchdiffeq.odeint(method='euler', t=[0,12]) (ode_demo.py:188) is restated as
    its fixed-grid semantics ``y1 = y0 + (t1-t0) * f(t0, y0)``.
"""
import math
from typing import Optional

import numpy as np
import torch
from torch import nn
from torch.nn import functional as F

OBLIQUE_EPS_F32 = 1e-4  # core/manifolds/oblique.py:7


# --------------------------------------------------------------------------
# Oblique manifold (core/manifolds/oblique.py:15-16, 36-45)
# --------------------------------------------------------------------------
def oblique_proj(p: torch.Tensor) -> torch.Tensor:
    """x / ||x||_2 over the last dim, no epsilon (oblique.py:15-16)."""
    return p / p.norm(dim=-1, keepdim=True)


def oblique_dist(p1: torch.Tensor, p2: torch.Tensor) -> torch.Tensor:
    """acos(clamp(p2 @ p1^T)) -> [..., rows(p2), rows(p1)] (oblique.py:36-43).

    Note the argument order: dist(q, k) returns a [S, L] matrix (rows = keys).
    """
    inner = p2 @ p1.transpose(-2, -1)
    inner = inner.clamp(-1 + OBLIQUE_EPS_F32, 1 - OBLIQUE_EPS_F32)
    return torch.acos(inner)


# --------------------------------------------------------------------------
# MHGSA (hyptransformerlib.py:29-311)
# --------------------------------------------------------------------------
def mhgsa(query, key, value, num_heads, in_w, in_b, out_w, out_b):
    """Multi-head geodesic attention.

    query [L, Nb, E], key/value [S, Nb, E] -> (out [L, Nb, E], head-mean weights [Nb, L, S]).
    Reproduces: packed in-proj (hyptransformerlib.py:113-168), q scaling (:191),
    head split (:214-218), scores = -Oblique.dist(proj(q), proj(k)) (:251-254),
    the transpose-only-if-shape-differs quirk (:261-265), softmax over the last
    dim (:294), bmm with v (:300), out_proj (:305), weights averaged over heads (:306-309).
    """
    L, Nb, E = query.shape
    S = key.shape[0]
    hd = E // num_heads
    q = F.linear(query, in_w[:E], in_b[:E])
    k = F.linear(key, in_w[E:2 * E], in_b[E:2 * E])
    v = F.linear(value, in_w[2 * E:], in_b[2 * E:])
    q = q * (float(hd) ** -0.5)
    q = q.contiguous().view(L, Nb * num_heads, hd).transpose(0, 1)
    k = k.contiguous().view(S, Nb * num_heads, hd).transpose(0, 1)
    v = v.contiguous().view(S, Nb * num_heads, hd).transpose(0, 1)
    w = -oblique_dist(oblique_proj(q), oblique_proj(k))  # [Nb*H, S, L]
    if list(w.shape) != [Nb * num_heads, L, S]:
        w = w.transpose(1, 2)
    w = torch.softmax(w, dim=-1)
    o = torch.bmm(w, v)  # requires w to be [*, L, S]; for L == S rows are keys (the quirk)
    o = o.transpose(0, 1).contiguous().view(L, Nb, E)
    o = F.linear(o, out_w, out_b)
    return o, w.view(Nb, num_heads, L, S).sum(dim=1) / num_heads


class HypMHSA(nn.Module):
    """Parameter holder with Hyp_mhsa's names (hyptransformerlib.py:314-381)."""

    def __init__(self, embed_dim, num_heads):
        super().__init__()
        self.embed_dim, self.num_heads = embed_dim, num_heads
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * embed_dim))
        self.out_proj = nn.Linear(embed_dim, embed_dim)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.zeros_(self.out_proj.bias)

    def forward(self, q, k, v):
        return mhgsa(q, k, v, self.num_heads, self.in_proj_weight, self.in_proj_bias,
                     self.out_proj.weight, self.out_proj.bias)


class HypAttention(nn.Module):
    """hypertransformer.py:19-89: MHGSA over dim0, then tanh(info) * sigmoid(gate)."""

    def __init__(self, d_model, nhead):
        super().__init__()
        self.model_dim = d_model
        self.temporal_attention_before = HypMHSA(d_model, nhead)
        self.temporal_info = nn.Linear(d_model, d_model)
        self.temporal_gate = nn.Linear(d_model, d_model)

    def forward(self, query, key, value):
        # [T, N, sample, D] -> [T, N*sample, D]  (hypertransformer.py:75-77)
        Lq, A, Sn, D = query.shape
        Lk = key.shape[0]
        a, w = self.temporal_attention_before(query.reshape(Lq, A * Sn, D), key.reshape(Lk, A * Sn, D),
                                              value.reshape(Lk, A * Sn, D))
        out = torch.tanh(self.temporal_info(a)) * torch.sigmoid(self.temporal_gate(a))
        return out.reshape(Lq, A, Sn, D), w


class EncoderLayer(nn.Module):
    """hypertransformer.py:91-153 (post-LN, relu FFN, dropout p=0)."""

    def __init__(self, d_model, nhead, ff):
        super().__init__()
        self.self_attn = HypAttention(d_model, nhead)
        self.linear1 = nn.Linear(d_model, ff)
        self.linear2 = nn.Linear(ff, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)

    def forward(self, src):
        src = self.norm1(src + self.self_attn(src, src, src)[0])
        src = self.norm2(src + self.linear2(F.relu(self.linear1(src))))
        return src


class _OdeFunc(nn.Module):
    def __init__(self, layer, nlayer):
        super().__init__()
        assert nlayer == 1  # model/STTODE.py:193 (nlayer = 1 is the only instantiated value)
        self.layers = nn.ModuleList([layer])

    def forward(self, t, x):
        for m in self.layers:
            x = m(x)
        return x


class _OdeBlock(nn.Module):
    def __init__(self, func, t1):
        super().__init__()
        self.odefunc = func
        self.t1 = float(t1)

    def forward(self, x):
        # torchdiffeq fixed-grid euler on t=[0, t1] with no step_size: ONE step (ode_demo.py:186-190)
        return x + self.t1 * self.odefunc(0.0, x)


class ODEGEncoder(nn.Module):
    """ode_demo.py:217-231: relu(x + T * Layer(x))."""

    def __init__(self, layer, nlayer, time):
        super().__init__()
        self.odeblock = _OdeBlock(_OdeFunc(layer, nlayer), time)

    def forward(self, src):
        return F.relu(self.odeblock(src))


class DecoderLayer(nn.Module):
    """hypertransformer.py:156-236 (never instantiated by STTODENet; op-level parity only): self-attention, cross-attention
    over a memory of a different length, relu FFN, three post-LayerNorms; every dropout is p = 0 in the configurations the
    repo builds, and the masks are accepted and ignored by Hypattention.forward (:55-89)."""

    def __init__(self, d_model, nhead, ff):
        super().__init__()
        self.self_attn = HypAttention(d_model, nhead)
        self.cross_attn = HypAttention(d_model, nhead)
        self.linear1 = nn.Linear(d_model, ff)
        self.linear2 = nn.Linear(ff, d_model)
        self.norm1, self.norm2, self.norm3 = nn.LayerNorm(d_model), nn.LayerNorm(d_model), nn.LayerNorm(d_model)

    def forward(self, tgt, memory):
        a, w_self = self.self_attn(tgt, tgt, tgt)
        tgt = self.norm1(tgt + a)
        a, w_cross = self.cross_attn(tgt, memory, memory)
        tgt = self.norm2(tgt + a)
        tgt = self.norm3(tgt + self.linear2(F.relu(self.linear1(tgt))))
        return tgt, w_self, w_cross


class ODEGDecoder(nn.Module):
    """ode_demo.py:195-213 ``ODEG`` over ``TransformerDecoder_ode`` (:74-133): relu(tgt + T * Stack(tgt, memory)), one Euler step."""

    def __init__(self, layers, time):
        super().__init__()
        self.layers = nn.ModuleList(layers)
        self.time = float(time)

    def forward(self, tgt, memory):
        x = tgt
        for m in self.layers:
            x = m(x, memory)[0]
        return F.relu(tgt + self.time * x)


def ode_integrate_ref(f, y0, t1, method='euler', steps=1):
    """Fixed-grid integration of y' = f(y) on a uniform grid of ``steps`` steps over [0, t1]: Euler, torchdiffeq's fixed-grid
    'rk4' (the 3/8-rule ``rk4_alt_step_func``) and the classical RK4.  Only the single Euler step is ever run by the reference
    (ode_demo.py:186-190); the others have no reference pin (torchdiffeq absent, SURVEY.md §8c: "parity unpinned")."""
    h = float(t1) / steps
    y = y0
    for _ in range(steps):
        k1 = f(y)
        if method == 'euler':
            y = y + h * k1
        elif method == 'rk4':
            k2 = f(y + h * k1 / 3)
            k3 = f(y + h * (k2 - k1 / 3))
            k4 = f(y + h * (k1 - k2 + k3))
            y = y + h * (k1 + 3 * (k2 + k3) + k4) / 8
        elif method == 'rk4_classic':
            k2 = f(y + h * k1 / 2)
            k3 = f(y + h * k2 / 2)
            k4 = f(y + h * k3)
            y = y + h * (k1 + 2 * k2 + 2 * k3 + k4) / 6
        else:
            raise ValueError(method)
    return y


def sinusoid_table(max_len, d_model):
    """model/STTODE.py:149-155."""
    pe = torch.zeros(max_len, d_model)
    pos = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
    div = torch.exp(torch.arange(0, d_model, 2).float() * (-np.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


class PosEnc(nn.Module):
    """model/STTODE.py:137-176 (concat=True; dropout is identity in eval)."""

    def __init__(self, d_model, max_t_len=200):
        super().__init__()
        self.fc = nn.Linear(2 * d_model, d_model)
        self.register_buffer('pe', sinusoid_table(max_t_len, d_model))

    drop_mask = None   # training-mode nn.Dropout(0.1) (model/STTODE.py:140,176) as an injected [n, T, D] mask (kept / 0.9 | 0)

    def forward(self, x):  # [n, T, D]
        pe = self.pe[: x.shape[1]][None].expand(x.shape[0], -1, -1)
        y = self.fc(torch.cat([x, pe], dim=-1))
        return y if self.drop_mask is None else y * self.drop_mask.view_as(y)


def add_category(x):
    """model/STTODE.py:199-210: append [0,0,0]; agent index N-1 gets [0,0,1]."""
    B, N = x.shape[0], x.shape[1]
    cat = torch.zeros(N, 3, dtype=x.dtype, device=x.device)
    cat[N - 1, 2] = 1
    return torch.cat((x, cat[None].expand(B, -1, -1)), dim=-1)


class _Trunk(nn.Module):
    """Shared trunk of PastEncoder / FutureEncoder (model/STTODE.py:214-236, 276-295)."""

    def __init__(self, args, length, in_dim=4):
        super().__init__()
        D = args.hidden_dim
        self.model_dim = D
        self.input_fc = nn.Linear(in_dim, D)
        self.input_fc2 = nn.Linear(D * length, D)
        self.input_fc3 = nn.Linear(D + 3, D)
        self.ODE_Encoder = ODEGEncoder(EncoderLayer(D, 8, 1024), 1, 12)
        self.pos_encoder = PosEnc(D)

    def trunk(self, inputs, batch_size, agent_num):
        T = inputs.shape[1]
        D = self.model_dim
        x = self.input_fc(inputs).view(batch_size * agent_num, T, D)
        x = self.pos_encoder(x).view(batch_size, agent_num, T * D)
        g = self.input_fc3(add_category(self.input_fc2(x)))  # [B, N, D]
        ode = self.ODE_Encoder(g.unsqueeze(2)).squeeze(2)
        return torch.cat((g, ode), dim=-1).view(batch_size * agent_num, -1), g, ode


class PastEncoder(_Trunk):
    def __init__(self, args):
        super().__init__(args, args.past_length)

    def forward(self, inputs, batch_size, agent_num):
        return self.trunk(inputs, batch_size, agent_num)[0]


class _AffineStack(nn.Module):  # MLP2 (model/STTODE.py:111-133) with relu
    def __init__(self, din, dims):
        super().__init__()
        self.affine_layers = nn.ModuleList()
        for d in dims:
            self.affine_layers.append(nn.Linear(din, d))
            din = d
        self.out_dim = din

    def forward(self, x):
        for a in self.affine_layers:
            x = torch.relu(a(x))
        return x


class FutureEncoder(_Trunk):
    """model/STTODE.py:238-300."""

    def __init__(self, args):
        super().__init__(args, args.future_length)
        scale_num = 2 + len(args.hyper_scales)
        self.out_mlp = _AffineStack(scale_num * args.hidden_dim, [128])
        self.qz_layer = nn.Linear(128, 2 * args.zdim)

    def forward(self, inputs, batch_size, agent_num, past_feature):
        ff = self.trunk(inputs, batch_size, agent_num)[0]
        return self.qz_layer(self.out_mlp(torch.cat((past_feature, ff), dim=-1)))


class _ReluMLP(nn.Module):  # model/utils.py:67-95
    def __init__(self, din, dout, hidden):
        super().__init__()
        dims = [din, *hidden, dout]
        self.layers = nn.ModuleList(nn.Linear(a, b) for a, b in zip(dims[:-1], dims[1:]))

    def forward(self, x):
        for i, l in enumerate(self.layers):
            x = l(x)
            if i != len(self.layers) - 1:
                x = torch.relu(x)
        return x


class DecomposeBlock(nn.Module):
    """model/STTODE.py:16-77."""

    def __init__(self, past_len, future_len, input_dim):
        super().__init__()
        self.past_len, self.future_len = past_len, future_len
        self.conv_past = nn.Conv1d(2, 32, 3, stride=1, padding=1)
        self.encoder_past = nn.GRU(32, 96, 1, batch_first=True)
        self.decoder_y = _ReluMLP(96 + input_dim, future_len * 2, (512, 256))
        self.decoder_x = _ReluMLP(96 + input_dim, past_len * 2, (512, 256))

    def forward(self, x_true, x_hat, f):
        e = torch.relu(self.conv_past((x_true - x_hat).transpose(1, 2))).transpose(1, 2)
        state = self.encoder_past(e)[1].squeeze(0)
        feat = torch.cat((f, state), dim=1)
        return (self.decoder_x(feat).view(-1, self.past_len, 2),
                self.decoder_y(feat).view(-1, self.future_len, 2), state)


class Decoder(nn.Module):
    """model/STTODE.py:302-347."""

    def __init__(self, args):
        super().__init__()
        self.past_length, self.future_length = args.past_length, args.future_length
        input_dim = 2 * args.hidden_dim + args.zdim
        self.decompose = nn.ModuleList(DecomposeBlock(args.past_length, args.future_length, input_dim)
                                       for _ in range(args.num_decompose))

    def forward(self, past_feature, z, past_traj, cur_location, sample_num, mode='train', trace=None):
        m = past_traj.shape[0] * sample_num
        hidden = torch.cat((past_feature.view(-1, sample_num, past_feature.shape[-1]),
                            z.view(-1, sample_num, z.shape[-1])), dim=-1).view(m, -1)
        x_true = past_traj.repeat_interleave(sample_num, dim=0)
        x_hat = torch.zeros_like(x_true)
        pred = torch.zeros(m, self.future_length, 2, dtype=x_true.dtype, device=x_true.device)
        rec = torch.zeros(m, self.past_length, 2, dtype=x_true.dtype, device=x_true.device)
        for i, blk in enumerate(self.decompose):
            x_hat, y_hat, state = blk(x_true, x_hat, hidden)
            if trace is not None:
                trace[f'x_hat{i}'], trace[f'y_hat{i}'], trace[f'state{i}'] = x_hat, y_hat, state
            pred = pred + y_hat
            rec = rec + x_hat
        out = pred + cur_location.repeat_interleave(sample_num, dim=0)
        if mode == 'inference':
            out = out.view(-1, sample_num, *out.shape[1:])
        return out, rec


class Normal:
    """model/STTODE.py:79-109."""

    def __init__(self, mu=None, logvar=None, params=None):
        if params is not None:
            mu, logvar = torch.chunk(params, 2, dim=-1)
        self.mu, self.logvar = mu, logvar
        self.sigma = torch.exp(0.5 * logvar)

    def rsample(self, eps=None):
        if eps is None:
            eps = torch.randn_like(self.sigma)
        return self.mu + eps * self.sigma

    def kl(self, p=None):
        if p is None:
            return -0.5 * (1 + self.logvar - self.mu.pow(2) - self.logvar.exp())
        t1 = (self.mu - p.mu) / (p.sigma + 1e-8)
        t2 = self.sigma / (p.sigma + 1e-8)
        return 0.5 * (t1 * t1 + t2 * t2) - 0.5 - torch.log(t2)


def first_diff_dup(traj):
    """Velocities with the first element duplicated (model/STTODE.py:432-433, 582-583). traj [n,T,2]."""
    v = traj[:, 1:] - traj[:, :-1]
    return torch.cat([v[:, :1], v], dim=1)


class STTODENetRef(nn.Module):
    """model/STTODE.py:349-623 restated (inference + forward-loss values)."""

    def __init__(self, args, device='cpu'):
        super().__init__()
        self.args, self.device = args, torch.device(device)
        scale_num = 2 + len(args.hyper_scales)
        self.past_encoder = PastEncoder(args)
        self.pz_layer = nn.Linear(scale_num * args.hidden_dim, 2 * args.zdim)
        self.future_encoder = FutureEncoder(args)
        self.decoder = Decoder(args)
        self.param_annealers = nn.ModuleList()

    # -- data entry ------------------------------------------------------
    def set_data(self, batch, pre_motion, fut_motion, pre_motion_mask=None, fut_motion_mask=None, theta=None):
        """model/STTODE.py:397-461 (eval branch: no rotation / subsampling unless theta given)."""
        self.batch_size = 1
        pre = pre_motion.permute(2, 0, 1)  # [Tp, N, 2]
        fut = fut_motion.permute(2, 0, 1)
        self.agent_num = pre.shape[1]
        self.scene_orig = pre[-1].reshape(-1, 2).mean(dim=0)
        if theta is not None:
            c, s = math.cos(theta), math.sin(theta)

            def rot(x):
                n = x - self.scene_orig
                r = torch.stack([n[..., 0] * c - n[..., 1] * s, n[..., 0] * s + n[..., 1] * c], dim=-1)
                return r + self.scene_orig, r
            pre, pre_n = rot(pre)
            fut, fut_n = rot(fut)
        else:
            pre_n, fut_n = pre - self.scene_orig, fut - self.scene_orig
        pre_vel = pre[1:] - pre[:-1]
        pre_vel = torch.cat([pre_vel[:1], pre_vel], dim=0)
        fut_vel = fut - torch.cat([pre[-1:], fut[:-1]])
        self.inputs = torch.cat([pre_n, pre_vel], dim=-1).permute(1, 0, 2)
        self.inputs_for_posterior = torch.cat([fut_n, fut_vel], dim=-1).permute(1, 0, 2)
        self.past_traj = pre_n.permute(1, 0, 2)
        self.future_traj = fut_n.permute(1, 0, 2)
        self.cur_location = self.past_traj[:, -1:]

    def set_data_nba(self, data):
        """model/STTODE.py:463-486."""
        a = self.args
        self.batch_size, self.agent_num = data['past_traj'].shape[:2]
        n = self.batch_size * self.agent_num
        self.past_traj = data['past_traj'].reshape(n, a.past_length, 2).contiguous()
        self.future_traj = data['future_traj'].reshape(n, a.future_length, 2).contiguous()
        self.scene_orig = self.past_traj
        self.cur_location = self.past_traj[:, -1:]
        self.inputs = torch.cat((self.past_traj, first_diff_dup(self.past_traj)), dim=-1)
        fv = self.future_traj - torch.cat([self.past_traj[:, -1:], self.future_traj[:, :-1]], dim=1)
        self.inputs_for_posterior = torch.cat((self.future_traj, fv), dim=-1)

    # -- compute entry ---------------------------------------------------
    @torch.no_grad()
    def inference(self, data=None, z=None, trace=None):
        """model/STTODE.py:574-623.  Returns [K, B*N, Tf, 2]."""
        a = self.args
        K = 20  # hard-coded sample_num (model/STTODE.py:600)
        if a.dataset == 'nba':
            B, N = data['past_traj'].shape[:2]
            past_traj = data['past_traj'].reshape(B * N, a.past_length, 2).contiguous()
        else:
            B, N, past_traj = 1, self.agent_num, self.past_traj
        inputs = torch.cat((past_traj, first_diff_dup(past_traj)), dim=-1)
        cur = past_traj[:, -1:]
        pf = self.past_encoder(inputs, B, N)
        pf_rep = pf.repeat_interleave(K, dim=0)
        if z is None:
            z = torch.randn(pf_rep.shape[0], a.zdim)
        if trace is not None:
            trace['past_feature'] = pf
        out, _ = self.decoder(pf_rep, z, past_traj, cur, sample_num=a.sample_k, mode='inference', trace=trace)
        out = out.permute(1, 0, 2, 3)
        if a.dataset != 'nba':
            out = out + self.scene_orig
        return out

    # -- staged API driven by the stage-2 sampler (sampler.py:36-60); differentiable (callers wrap in no_grad for values) --
    def encode_history(self):
        """model/STTODE.py:488-496."""
        self.past_feature = self.past_encoder(self.inputs, self.batch_size, self.agent_num)

    def decoder_future_0(self, z, eps20=None):
        """model/STTODE.py:534-551: K=1 decode, then the N(0,I) prior over the 20 samples (drawn, stored as pz_dis)."""
        a = self.args
        self.pred_traj, self.recover_traj = self.decoder(self.past_feature, z, self.past_traj, self.cur_location, sample_num=1)
        m = self.past_feature.shape[0] * 20
        self.pz_dis = Normal(mu=torch.zeros(m, a.zdim), logvar=torch.zeros(m, a.zdim))
        self.pz_sampled = self.pz_dis.rsample(eps20)

    def decoder_future_1(self, pz):
        """model/STTODE.py:529-532."""
        self.diverse_pred_traj, _ = self.decoder(self.past_feature.repeat_interleave(20, dim=0), pz, self.past_traj,
                                                 self.cur_location, sample_num=20, mode='inference')

    @torch.no_grad()
    def forward_losses(self, eps_q, eps_p1, eps_p20):
        """model/STTODE.py:553-568 with injected noises; returns the five loss values as floats."""
        return tuple(float(v) for v in self.forward_loss_tensors(eps_q, eps_p1, eps_p20))

    def forward_loss_tensors(self, eps_q, eps_p1, eps_p20):
        """Same, as tensors and WITHOUT no_grad: ``[0].backward()`` is the autograd reference for the HIP training step
        (what train.py:83-85 does).  Dropout masks: set ``{past,future}_encoder.pos_encoder.drop_mask``."""
        a = self.args
        B, N = self.batch_size, self.agent_num
        pf = self.past_encoder(self.inputs, B, N)
        qz = Normal(params=self.future_encoder(self.inputs_for_posterior, B, N, pf))
        qz_s = qz.rsample(eps_q)
        pz = Normal(mu=torch.zeros(pf.shape[0], a.zdim), logvar=torch.zeros(pf.shape[0], a.zdim))
        pz.rsample(eps_p1)  # drawn and overwritten by the reference (model/STTODE.py:525,549)
        pred, rec = self.decoder(pf, qz_s, self.past_traj, self.cur_location, sample_num=1)
        pz20 = eps_p20  # prior N(0, I): mu + eps * 1
        lp = (self.future_traj - pred).pow(2).sum() / B / pred.shape[1]
        lr = (self.past_traj - rec).pow(2).sum() / B / rec.shape[1]
        lk = (qz.kl(pz).sum() / (B * N)).clamp_min(a.min_clip)
        div, _ = self.decoder(pf.repeat_interleave(20, dim=0), pz20, self.past_traj, self.cur_location,
                              sample_num=20, mode='inference')
        ld = (self.future_traj.unsqueeze(1) - div).pow(2).sum(-1).sum(-1).min(dim=1)[0].mean()
        return lp + lr + lk + ld, lp, lr, lk, ld
