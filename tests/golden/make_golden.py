#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING THE REFERENCE (/root/reference).

Run in the authoring container only (the reference does not travel to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What is stored is data only: seeded inputs, the injected latents and the reference's outputs.
Weights are NOT stored; they are regenerated from sttode_amd.weights.make_weights(seed) and loaded
into the reference with load_state_dict(strict=True).

The reference needs five shims to import on CPU / torch 2.x; they are applied here,
outside the reference tree:
  1. stub module ``glob2`` (imported, never used: model/utils.py:8);
  2. stub module ``torchdiffeq`` whose odeint is the library's documented fixed-grid Euler
     (grid = t when no step_size; y1 = y0 + dt * f(t0, y0)) -- torchdiffeq==0.2.3 is not installed
     (requirement.txt:195).  This boundary is pinned only by our own fixture (reference has no test for it);
  3. torch.nn.modules.linear._LinearWithBias = Linear (hypertransformer.py:11, never used);
  4. Tensor.cuda = identity (model/STTODE.py:333-334);
  5. torch.zeros(device='cuda') -> cpu (hypertransformer.py:69).
Randomness (torch.randn_like, model/STTODE.py:92) is replaced by a queue of pre-seeded tensors so
the latents are fixture inputs.
"""
import argparse
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = os.environ.get('STTODE_REFERENCE', '/root/reference')


def install_shims():
    sys.modules['glob2'] = types.ModuleType('glob2')
    tde = types.ModuleType('torchdiffeq')

    def odeint(func, y0, t, method='euler', **kw):
        assert method == 'euler' and not kw
        ys = [y0]
        for i in range(len(t) - 1):
            ys.append(ys[-1] + (t[i + 1] - t[i]) * func(t[i], ys[-1]))
        return torch.stack(ys)
    tde.odeint = tde.odeint_adjoint = odeint
    sys.modules['torchdiffeq'] = tde
    import torch.nn.modules.linear as L
    L._LinearWithBias = L.Linear
    torch.Tensor.cuda = lambda self, *a, **k: self
    _zeros = torch.zeros

    def zeros(*a, **k):
        if k.get('device') == 'cuda':
            k['device'] = 'cpu'
        return _zeros(*a, **k)
    torch.zeros = zeros
    if REF not in sys.path:
        sys.path.insert(0, REF)


class NoiseQueue:
    """Replaces torch.randn_like: pops pre-seeded tensors (shape-checked)."""

    def __init__(self):
        self.q = []
        self._orig = torch.randn_like
        torch.randn_like = self

    def push(self, *arrs):
        self.q.extend(torch.from_numpy(np.asarray(a)) for a in arrs)

    def __call__(self, like, **kw):
        t = self.q.pop(0)
        assert tuple(t.shape) == tuple(like.shape), (t.shape, like.shape)
        return t.to(like.dtype)


def make_args(dataset, Tp, Tf):
    return argparse.Namespace(hidden_dim=64, zdim=32, hyper_scales=[5, 11], num_decompose=2, past_length=Tp,
                              future_length=Tf, sample_k=20, learn_prior=False, ztype='gaussian', dataset=dataset,
                              min_clip=2.0, max_train_agent=32, rand_rot_scene=True, discrete_rot=False)


def build_ref(dataset, Tp, Tf, seed=1234):
    from model.STTODE import STTODENet
    from sttode_amd.weights import make_weights, to_torch_state_dict
    m = STTODENet(make_args(dataset, Tp, Tf), torch.device('cpu')).eval()
    m.load_state_dict(to_torch_state_dict(make_weights(seed, past_length=Tp, future_length=Tf)), strict=True)
    return m


def capture(module, store, key):
    def hook(_m, _inp, out):
        store[key] = [o.detach().clone() for o in out] if isinstance(out, (tuple, list)) else out.detach().clone()
    return module.register_forward_hook(hook)


def npy(x):
    return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)


def eth_like_case(m, noise, obs, pred, zseed):
    """One scene through set_data + inference (test.py:171-199 flow)."""
    from sttode_amd.scenes import latents
    from utils.metrics import compute_ADE, compute_FDE
    N = obs.shape[0]
    z = latents(zseed, N)
    cap = {}
    hs = [capture(m.past_encoder, cap, 'pf'), capture(m.decoder.decompose[0], cap, 'b0'),
          capture(m.decoder.decompose[1], cap, 'b1')]
    with torch.no_grad():
        o, p = torch.from_numpy(obs), torch.from_numpy(pred)
        m.set_data(None, o, p, torch.ones(N, obs.shape[2]), torch.ones(N, pred.shape[2]))
        noise.push(z)
        out = m.inference(None)  # [K, N, Tf, 2]
    for h in hs:
        h.remove()
    dec = out.permute(1, 0, 2, 3).numpy()
    gt = p.transpose(1, 2).numpy()
    ade = compute_ADE([dec[i] for i in range(N)], gt)
    fde = compute_FDE([dec[i] for i in range(N)], gt)
    return dict(obs=obs, pred=pred, z=z, past_feature=npy(cap['pf']), x_hat0=npy(cap['b0'][0]), y_hat0=npy(cap['b0'][1]),
                x_hat1=npy(cap['b1'][0]), y_hat1=npy(cap['b1'][1]), scene_orig=npy(m.scene_orig), out=npy(out),
                ade=np.float64(ade), fde=np.float64(fde))


def sampler_args(dataset, Tp, Tf):
    a = make_args(dataset, Tp, Tf)
    a.nz, a.qnet_mlp, a.share_eps, a.train_w_mean = 32, [512, 256], True, True
    a.kld_weight, a.kld_min_clamp, a.recon_weight = 0.1, 10.0, 5.0
    return a


def sampler_cases(noise):
    """Stage-2 Sampler (sampler.py:32-70) + its objective (samplerloss.py:41-73) on an ETH scene and an NBA batch,
    mean and sampled latent codes (shared and per-agent eps).  torch.randn (sampler.py:42,45) is replaced by the fixture eps."""
    from sampler import Sampler
    from samplerloss import compute_sampler_loss, compute_sampler_loss_nba
    from sttode_amd import scenes
    from sttode_amd.weights import make_sampler_weights, to_torch_state_dict
    out = {}
    _randn = torch.randn
    # NB the reference sizes eps by net.agent_num (agents PER SCENE, sampler.py:43,45), so for NBA the sampled modes only
    # work with one scene per forward call (B = 1); mean mode takes any B.
    for tag, dataset, Tp, Tf, B, modes in (('eth', 'eth', 8, 12, 1, ('mean', 'shared', 'peragent')),
                                           ('nba', 'nba', 5, 10, 6, ('mean',)), ('nba1', 'nba', 5, 10, 1, ('shared', 'peragent'))):
        net = build_ref(dataset, Tp, Tf)
        a = sampler_args(dataset, Tp, Tf)
        smp = Sampler(a).eval()
        smp.load_state_dict(to_torch_state_dict(make_sampler_weights()), strict=True)
        if dataset == 'eth':
            o, p = scenes.eth_scene(5011, n_min=11, n_max=11)
            n = 11
            setd = lambda: net.set_data(None, torch.from_numpy(o), torch.from_numpy(p), torch.ones(n, Tp), torch.ones(n, Tf))
            fut = torch.from_numpy(p).transpose(1, 2)
            out['eth_obs'], out['eth_pred'] = o, p
            div_cfg = {'weight': 1, 'scale': 1}       # trainsampler.py:102-116
        else:
            d = scenes.nba_batch(6, B)
            n = B * 11
            data = {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in d.items()}
            setd = lambda: net.set_data_nba(data)
            fut = data['future_traj'].reshape(n, Tf, 2)
            out[tag + '_seed'], out[tag + '_B'] = np.int64(6), np.int64(B)
            div_cfg = {'weight': 1, 'scale': 1.0}
        rng = np.random.default_rng(808 + n)
        for mode in modes:
            smp.share_eps = mode != 'peragent'
            eps = rng.standard_normal((1, 32) if smp.share_eps else (n, 32)).astype(np.float32)
            e_q, e_p, e20 = (rng.standard_normal(sh).astype(np.float32) for sh in ((n, 32), (n, 32), (n * 20, 32)))
            torch.randn = lambda *a_, **k_: torch.from_numpy(eps)
            with torch.no_grad():
                setd()
                noise.push(e_q, e_p, e20)
                dec, sd, vd, aw = smp.forward(net, mean=(mode == 'mean'))
                if dataset == 'nba':   # trainsampler.py:142-146
                    tot, ld, _ = compute_sampler_loss_nba(a, fut, dec.reshape(-1, 20, Tf, 2), 1, vd, sd, div_cfg)
                else:                  # trainsampler.py:176-180
                    tot, ld, _ = compute_sampler_loss(a, fut, dec, 1, torch.ones(n, Tf), vd, sd, div_cfg)
            torch.randn = _randn
            k = f'{tag}_{mode}_'
            out.update({k + 'eps': eps, k + 'dec': npy(dec), k + 'mu': npy(sd.mu), k + 'logvar': npy(sd.logvar),
                        k + 'pred_traj': npy(aw), k + 'loss': np.array([float(tot), float(ld['kld']), float(ld['diverse'])], np.float64)})
    np.savez(os.path.join(HERE, 'sampler.npz'), **out)
    print('sampler.npz bytes:', os.path.getsize(os.path.join(HERE, 'sampler.npz')))


def grad_summary(t):
    """Per-parameter gradient digest kept in the fixture: [sum, L2 norm, max |g|] + the first 48 entries."""
    g = t.detach().double().flatten()
    return np.concatenate([[g.sum().item(), g.norm().item(), g.abs().max().item()], g[:48].numpy()])


def grad_cases(noise):
    """total_loss.backward() of the reference (train.py:83-85) with injected noises, module in eval() so that the random
    rotation / sub-sampling of set_data (model/STTODE.py:405-426) and the positional dropout are off: ETH N=7 and NBA B=4."""
    from sttode_amd import scenes
    out = {}
    for tag, dataset, Tp, Tf in (('eth', 'eth', 8, 12), ('nba', 'nba', 5, 10)):
        m = build_ref(dataset, Tp, Tf)
        m.zero_grad()
        if dataset == 'eth':
            o, p = scenes.eth_scene(5007, n_min=7, n_max=7)
            n = 7
            m.set_data(None, torch.from_numpy(o), torch.from_numpy(p), torch.ones(n, Tp), torch.ones(n, Tf))
            out['eth_obs'], out['eth_pred'] = o, p
        else:
            d = scenes.nba_batch(9, 4)
            n = 44
            m.set_data_nba({k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in d.items()})
            out['nba_seed'], out['nba_B'] = np.int64(9), np.int64(4)
        rng = np.random.default_rng(1300 + n)
        eq, ep1, ep20 = (rng.standard_normal(s).astype(np.float32) for s in ((n, 32), (n, 32), (n * 20, 32)))
        noise.push(eq, ep1, ep20)
        tot, lp, lr, lk, ld = m.forward()
        tot.backward()
        out.update({f'{tag}_eps_q': eq, f'{tag}_eps_p1': ep1, f'{tag}_eps_p20': ep20,
                    f'{tag}_losses': np.array([float(tot), lp, lr, lk, ld], np.float64)})
        for name, prm in m.named_parameters():
            if prm.grad is not None:
                out[f'{tag}_grad::{name}'] = grad_summary(prm.grad)
            else:
                out[f'{tag}_nograd::{name}'] = np.zeros(0)
    np.savez(os.path.join(HERE, 'forward_grads.npz'), **out)
    print('forward_grads.npz bytes:', os.path.getsize(os.path.join(HERE, 'forward_grads.npz')))


def sampler_grad_cases(noise):
    """Stage-2 training step of the reference (trainsampler.py:134-150,171-185): gradients of the Sampler's parameters for the
    'shared' ETH case and the 'mean' NBA case of sampler.npz (same inputs / eps); net in eval() (no rotation / dropout)."""
    from sampler import Sampler
    from samplerloss import compute_sampler_loss, compute_sampler_loss_nba
    from sttode_amd import scenes
    from sttode_amd.weights import make_sampler_weights, to_torch_state_dict
    g = dict(np.load(os.path.join(HERE, 'sampler.npz')))
    out = {}
    _randn = torch.randn
    for tag, dataset, Tp, Tf, mode in (('eth', 'eth', 8, 12, 'shared'), ('nba', 'nba', 5, 10, 'mean')):
        net = build_ref(dataset, Tp, Tf)
        a = sampler_args(dataset, Tp, Tf)
        smp = Sampler(a)
        smp.load_state_dict(to_torch_state_dict(make_sampler_weights()), strict=True)
        if dataset == 'eth':
            o, p = g['eth_obs'], g['eth_pred']
            n = o.shape[0]
            net.set_data(None, torch.from_numpy(o), torch.from_numpy(p), torch.ones(n, Tp), torch.ones(n, Tf))
            fut = torch.from_numpy(p).transpose(1, 2)
        else:
            d = scenes.nba_batch(int(g['nba_seed']), int(g['nba_B']))
            data = {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in d.items()}
            n = data['past_traj'].shape[0] * 11
            net.set_data_nba(data)
            fut = data['future_traj'].reshape(n, Tf, 2)
        rng = np.random.default_rng(99 + n)
        eps = g[f'{tag}_{mode}_eps']
        e_q, e_p, e20 = (rng.standard_normal(sh).astype(np.float32) for sh in ((n, 32), (n, 32), (n * 20, 32)))
        torch.randn = lambda *a_, **k_: torch.from_numpy(eps)
        noise.push(e_q, e_p, e20)
        dec, sd, vd, _ = smp.forward(net, mean=(mode == 'mean'))
        torch.randn = _randn
        cfg = {'weight': 1, 'scale': 1.0}
        if dataset == 'nba':
            tot, ld, _ = compute_sampler_loss_nba(a, fut, dec.reshape(-1, 20, Tf, 2), 1, vd, sd, cfg)
        else:
            tot, ld, _ = compute_sampler_loss(a, fut, dec, 1, torch.ones(n, Tf), vd, sd, cfg)
        tot.backward()
        out[f'{tag}_mode'] = np.array(mode)
        out[f'{tag}_loss'] = np.array([float(tot.detach()), float(ld['kld'].detach()), float(ld['diverse'].detach())], np.float64)
        for name, prm in smp.named_parameters():
            if prm.grad is not None:
                out[f'{tag}_grad::{name}'] = grad_summary(prm.grad)
            else:
                out[f'{tag}_nograd::{name}'] = np.zeros(0)
    np.savez(os.path.join(HERE, 'sampler_grads.npz'), **out)
    print('sampler_grads.npz bytes:', os.path.getsize(os.path.join(HERE, 'sampler_grads.npz')))


def decoder_stack_cases():
    """The repo's unused decoder-side stack (hypertransformer.py:156-236, ode_demo.py:74-133,195-213): one
    TransformerDecoderLayer and ODEG (2 layers, time 3) with cross-attention over a memory of a different length."""
    from hypertransformer import TransformerDecoderLayer
    from ode_demo import ODEG
    from sttode_amd.weights import make_decoder_layer_weights, to_torch_state_dict
    rng = np.random.default_rng(62)
    out = {}
    layer = TransformerDecoderLayer(64, 8, 256, dropout=0.0).eval()
    layer.load_state_dict(to_torch_state_dict(make_decoder_layer_weights(61, d=64, ff=256)), strict=True)   # weights: recipe, not stored
    names = [k for k, _ in layer.named_parameters()]
    tgt = rng.standard_normal((6, 5, 2, 64)).astype(np.float32)
    mem = rng.standard_normal((9, 5, 2, 64)).astype(np.float32)
    with torch.no_grad():
        y, ws, wc = layer(torch.from_numpy(tgt), torch.from_numpy(mem), seq_mask=True, need_weights=True)
        ode = ODEG(layer, 2, 3).eval()                      # _get_clones: both layers start as copies of ``layer``
        z, _ = ode(torch.from_numpy(tgt), torch.from_numpy(mem), seq_mask=True)
    out.update(tgt=tgt, mem=mem, layer_out=npy(y), odeg_out=npy(z))
    np.savez(os.path.join(HERE, 'decoder_stack.npz'), **out)
    print('decoder_stack.npz bytes:', os.path.getsize(os.path.join(HERE, 'decoder_stack.npz')), len(names), 'tensors')


def pmath_grad_cases():
    """Backward rules of the library's custom autograd functions (hyptorch/pmath.py:16-60), through the reference's own classes."""
    import hyptorch.pmath as pm
    rng = np.random.default_rng(15)
    x = np.concatenate([np.linspace(-1.2, 1.2, 49), [0.99999, -0.99999, 0.999995, 3.0]]).astype(np.float32)
    g = rng.standard_normal(x.shape).astype(np.float32)
    out = {'x': x, 'g': g}
    for name, fn in (('artanh', pm.artanh), ('arsinh', lambda t: pm.arsinh(t * 20))):
        t = torch.from_numpy(x).clone().requires_grad_(True)
        fn(t).backward(torch.from_numpy(g))
        out[name + '_grad'] = npy(t.grad)
    xr = (rng.standard_normal((17, 16)) * 0.2).astype(np.float32)
    gr = rng.standard_normal((17, 16)).astype(np.float32)
    for c in (1.0, 0.5):
        pm.RiemannianGradient.c = c
        t = torch.from_numpy(xr).clone().requires_grad_(True)
        pm.RiemannianGradient.apply(t).backward(torch.from_numpy(gr))
        out[f'riem_c{c}_grad'] = npy(t.grad)
    pm.RiemannianGradient.c = 1
    out.update(xr=xr, gr=gr)
    np.savez(os.path.join(HERE, 'pmath_grads.npz'), **out)


def main():
    install_shims()
    noise = NoiseQueue()
    if '--only-pmath-grads' in sys.argv:
        pmath_grad_cases()
        return
    if '--only-decoder-stack' in sys.argv:
        decoder_stack_cases()
        return
    if '--only-sampler-grads' in sys.argv:
        sampler_grad_cases(noise)
        assert not noise.q
        return
    if '--only-grads' in sys.argv:
        grad_cases(noise)
        assert not noise.q
        return
    if '--only-sampler' in sys.argv:
        sampler_cases(noise)
        assert not noise.q
        return
    from sttode_amd import scenes
    from sttode_amd.weights import make_weights

    # ---------------- ETH-shaped scenes, N in {2,7,32} -------------------------------------------------
    m = build_ref('eth', 8, 12)
    # informational: recipe pe table vs the reference's own buffer
    from model.STTODE import PositionalAgentEncoding
    pe_ref = PositionalAgentEncoding(64).pe.numpy()
    print('pe table max|recipe - reference| =', np.abs(pe_ref - make_weights(1234)['past_encoder.pos_encoder.pe']).max())
    for N in (2, 7, 32):
        rng = np.random.default_rng(100 + N)
        o, p = scenes.eth_scene(5000 + N, n_min=N, n_max=N)
        np.savez(os.path.join(HERE, f'eth_N{N}.npz'), **eth_like_case(m, noise, o, p, zseed=N))
    # training-mode objective (model/STTODE.py:553-568) in eval mode (no rotation/subsample/dropout), injected noise
    o, p = scenes.eth_scene(5007, n_min=7, n_max=7)
    rng = np.random.default_rng(77)
    eq, ep1, ep20 = (rng.standard_normal(s).astype(np.float32) for s in ((7, 32), (7, 32), (140, 32)))
    with torch.no_grad():
        m.set_data(None, torch.from_numpy(o), torch.from_numpy(p), torch.ones(7, 8), torch.ones(7, 12))
        noise.push(eq, ep1, ep20)
        tot, lp, lr, lk, ld = m.forward()
    np.savez(os.path.join(HERE, 'eth_forward_losses.npz'), obs=o, pred=p, eps_q=eq, eps_p1=ep1, eps_p20=ep20,
             losses=np.array([float(tot), lp, lr, lk, ld], np.float64), pred_traj=npy(m.pred_traj),
             recover_traj=npy(m.recover_traj), diverse_pred_traj=npy(m.diverse_pred_traj), qz_param=npy(m.qz_param))

    # ---------------- SDD-shaped ragged scenes N in {1,3,17,40} ----------------------------------------
    sdd = {}
    for i, N in enumerate((1, 3, 17, 40)):
        rng = np.random.default_rng(SDD_SEED + N)
        pos = (np.around(rng.uniform(0, 1, (N, 1, 2)) * [1400, 1900] + rng.normal(0, 12, (N, 1, 2)) * np.arange(20)[None, :, None]
                         + rng.normal(0, 1, (N, 20, 2)), 2) / 50.0).astype(np.float32).transpose(0, 2, 1)
        c = eth_like_case(m, noise, np.ascontiguousarray(pos[:, :, :8]), np.ascontiguousarray(pos[:, :, 8:]), zseed=900 + N)
        for k in ('obs', 'pred', 'z', 'out', 'past_feature', 'ade', 'fde'):
            sdd[f's{i}_{k}'] = c[k]
    np.savez(os.path.join(HERE, 'sdd_ragged.npz'), **sdd)

    # ---------------- NBA: batch-as-sequence attention (B in {4,32,128}), Tp=5 Tf=10 N=11 --------------
    mn = build_ref('nba', 5, 10)
    for B in (4, 32, 128):
        d = scenes.nba_batch(B, B)
        z = scenes.latents(4000 + B, B * 11)
        data = {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in d.items()}
        cap = {}
        h = capture(mn.past_encoder, cap, 'pf')
        att = mn.past_encoder.ODE_Encoder.odeblock.odefunc.layers[0].self_attn.temporal_attention_before
        h2 = capture(att, cap, 'att')
        with torch.no_grad():
            mn.set_data_nba(data)
            noise.push(z)
            out = mn.inference(data)  # [20, B*11, 10, 2]
        h.remove(); h2.remove()
        sub = slice(None) if B <= 32 else slice(0, None, 16)  # keep the B=128 fixture small
        # inputs and z are regenerated from seeds by the tests (scenes.nba_batch(B, B), scenes.latents(4000 + B, B * 11))
        np.savez(os.path.join(HERE, f'nba_B{B}.npz'), nba_seed=np.int64(B), z_seed=np.int64(4000 + B),
                 past_feature=npy(cap['pf'])[sub], out=npy(out)[:, sub], attn_out=npy(cap['att'][0])[:, :, :8],
                 stride=np.int64(1 if B <= 32 else 16))
    # ---------------- long horizon: Tp=10 Tf=40 N=10 B=8 (BASELINE config 5 shapes) -------------------
    ml = build_ref('nba', 10, 40)
    d = scenes.nba_batch(8, 8, N=10, obs_len=10, pred_len=40)
    z = scenes.latents(4100, 80)
    data = {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in d.items()}
    with torch.no_grad():
        ml.set_data_nba(data)
        noise.push(z)
        out = ml.inference(data)
    np.savez(os.path.join(HERE, 'nba_long_B8.npz'), nba_seed=np.int64(8), z_seed=np.int64(4100), out=npy(out))

    # ---------------- op level: Hyp_mhsa (self L in {1,6,128}; cross L=6,S=9), ODEG_Encoder -----------
    from hyptransformerlib import Hyp_mhsa
    from core.manifolds import Oblique
    rng = np.random.default_rng(31)
    att = Hyp_mhsa(64, 8).eval()
    sd = {'in_proj_weight': rng.uniform(-0.3, 0.3, (192, 64)), 'in_proj_bias': 0.05 * rng.standard_normal(192),
          'out_proj.weight': rng.uniform(-0.2, 0.2, (64, 64)), 'out_proj.bias': 0.05 * rng.standard_normal(64)}
    sd = {k: torch.from_numpy(v.astype(np.float32)) for k, v in sd.items()}
    att.load_state_dict(sd)
    ops = {f'w_{k}': npy(v) for k, v in sd.items()}
    with torch.no_grad():
        for L in (1, 6, 128):
            x = torch.from_numpy(rng.standard_normal((L, 5, 64)).astype(np.float32))
            o, w = att(x, x, x)
            ops[f'self{L}_x'], ops[f'self{L}_out'], ops[f'self{L}_w'] = npy(x), npy(o), npy(w)
        q = torch.from_numpy(rng.standard_normal((6, 5, 64)).astype(np.float32))
        kv = torch.from_numpy(rng.standard_normal((9, 5, 64)).astype(np.float32))
        o, w = att(q, kv, kv)
        ops['cross_q'], ops['cross_kv'], ops['cross_out'], ops['cross_w'] = npy(q), npy(kv), npy(o), npy(w)
        M = Oblique()
        a, b = torch.from_numpy(rng.standard_normal((3, 7, 8)).astype(np.float32)), torch.from_numpy(rng.standard_normal((3, 5, 8)).astype(np.float32))
        ops['obl_a'], ops['obl_b'] = npy(a), npy(b)
        ops['obl_proj_a'] = npy(M.proj(a))
        ops['obl_dist'] = npy(M.dist(M.proj(a), M.proj(b)))  # [3, 5, 7]
        x = torch.from_numpy(rng.standard_normal((4, 11, 1, 64)).astype(np.float32))
        m.past_encoder.get_agent_mask(torch.zeros(1))
        ops['ode_x'], ops['ode_out'] = npy(x), npy(m.past_encoder.ODE_Encoder(x, mask=None, num_agent=11))
    np.savez(os.path.join(HERE, 'ops.npz'), **ops)

    # ---------------- pmath primitives ------------------------------------------------------------------
    import hyptorch.pmath as pm
    rng = np.random.default_rng(5)
    P = {}
    for c in (1.0, 0.5):
        tag = f'c{c}'
        x = torch.from_numpy((rng.standard_normal((33, 16)) * 0.18).astype(np.float32))
        y = torch.from_numpy((rng.standard_normal((33, 16)) * 0.18).astype(np.float32))
        x[0] = 0; y[1] = 0; x[2] = x[2] / x[2].norm() * 2.0  # zero rows and an out-of-ball row (projection branch)
        u = torch.from_numpy((rng.standard_normal((33, 16)) * 0.5).astype(np.float32)); u[3] = 0
        mat = torch.from_numpy((rng.standard_normal((12, 16)) * 0.4).astype(np.float32))
        xb = pm.project(x, c=c)
        yb = pm.project(y, c=c)
        P.update({f'{tag}_x': npy(x), f'{tag}_y': npy(y), f'{tag}_u': npy(u), f'{tag}_m': npy(mat),
                  f'{tag}_project': npy(xb), f'{tag}_lambda_x': npy(pm.lambda_x(xb, c=c)),
                  f'{tag}_mobius_add': npy(pm.mobius_add(xb, yb, c=c)), f'{tag}_dist': npy(pm.dist(xb, yb, c=c)),
                  f'{tag}_dist0': npy(pm.dist0(xb, c=c)), f'{tag}_expmap': npy(pm.expmap(xb, u, c=c)),
                  f'{tag}_expmap0': npy(pm.expmap0(u, c=c)), f'{tag}_logmap': npy(pm.logmap(xb, yb, c=c)),
                  f'{tag}_logmap0': npy(pm.logmap0(xb, c=c)), f'{tag}_mobius_matvec': npy(pm.mobius_matvec(mat, xb, c=c)),
                  f'{tag}_p2k': npy(pm.p2k(xb, c)), f'{tag}_k2p': npy(pm.k2p(pm.p2k(xb, c), c)),
                  f'{tag}_lorenz': npy(pm.lorenz_factor(pm.p2k(xb, c), c=c)),
                  f'{tag}_poincare_mean': npy(pm.poincare_mean(xb, dim=0, c=c)),
                  f'{tag}_dist_matrix': npy(pm.dist_matrix(xb, yb[:9], c=c)),
                  f'{tag}_mobius_addition_batch': npy(pm._mobius_addition_batch(xb[:6], yb[:5], torch.tensor(c))),
                  f'{tag}_hyperbolic_softmax': npy(pm._hyperbolic_softmax(xb, mat * 0.5, pm.project(mat * 0.3, c=c), torch.tensor(c)))})
    s = torch.from_numpy(np.concatenate([np.linspace(-1.5, 1.5, 61), [20.0, -20.0, 0.999999, -0.999999]]).astype(np.float32))
    P.update(scalar_in=npy(s), tanh=npy(pm.tanh(s)), artanh=npy(pm.artanh(s)), arsinh=npy(pm.arsinh(s * 30)))
    P['auto_select_c'] = np.array([pm.auto_select_c(d) for d in (2, 8, 16, 64)], np.float64)
    np.savez(os.path.join(HERE, 'pmath.npz'), **P)

    # ---------------- metrics (utils/metrics.py) ---------------------------------------------------------
    from utils.metrics import compute_ADE, compute_FDE
    rng = np.random.default_rng(9)
    pr = rng.standard_normal((13, 20, 12, 2)).astype(np.float32)
    gt = rng.standard_normal((13, 12, 2)).astype(np.float32)
    np.savez(os.path.join(HERE, 'metrics.npz'), pred=pr, gt=gt, ade=np.float64(compute_ADE(list(pr), gt)),
             fde=np.float64(compute_FDE(list(pr), gt)))
    sampler_cases(noise)
    grad_cases(noise)
    sampler_grad_cases(noise)
    decoder_stack_cases()
    pmath_grad_cases()
    assert not noise.q
    tot = sum(os.path.getsize(os.path.join(HERE, f)) for f in os.listdir(HERE) if f.endswith('.npz'))
    print('golden bytes:', tot)


SDD_SEED = 770000

if __name__ == '__main__':
    main()
