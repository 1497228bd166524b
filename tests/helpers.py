"""Shared test helpers: argument namespaces, oracle construction, tolerances."""
import argparse

import numpy as np
import torch

# Parity bar (BASELINE.json north_star): 1e-4 relative on predicted coordinates, fp32.
# Outputs near zero make a pure relative test meaningless, so: |a-b| <= ATOL + RTOL*|b|.
RTOL = 1e-4
ATOL = 1e-4
# oracle-vs-reference (same PyTorch CPU ops, same order): much tighter
ORACLE_ATOL = 2e-5


def make_args(dataset='eth', Tp=8, Tf=12):
    return argparse.Namespace(hidden_dim=64, zdim=32, hyper_scales=[5, 11], num_decompose=2, past_length=Tp,
                              future_length=Tf, sample_k=20, learn_prior=False, ztype='gaussian', dataset=dataset,
                              min_clip=2.0, max_train_agent=32, rand_rot_scene=True, discrete_rot=False)


_ORACLES = {}


def oracle_model(dataset='eth', Tp=8, Tf=12, seed=1234):
    from oracle.sttode_ref import STTODENetRef
    from sttode_amd.weights import make_weights, to_torch_state_dict
    key = (dataset, Tp, Tf, seed)
    if key not in _ORACLES:
        m = STTODENetRef(make_args(dataset, Tp, Tf)).eval()
        m.load_state_dict(to_torch_state_dict(make_weights(seed, past_length=Tp, future_length=Tf)), strict=True)
        _ORACLES[key] = m
    return _ORACLES[key]


def oracle_scene_inference(m, obs, pred, z, trace=None):
    """test.py:171-186 flow on the oracle: returns [K, N, Tf, 2]."""
    with torch.no_grad():
        m.set_data(None, torch.from_numpy(obs), torch.from_numpy(pred))
        return m.inference(None, z=torch.from_numpy(z), trace=trace).numpy()


def assert_close(a, b, rtol=RTOL, atol=ATOL, what=''):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = np.abs(a - b)
    bound = atol + rtol * np.abs(b)
    bad = err > bound
    assert not bad.any(), f'{what}: {bad.sum()} / {bad.size} out of tolerance, max err {err.max():.3e} (max |ref| {np.abs(b).max():.3e})'


def sampler_args(dataset='eth', Tp=8, Tf=12):
    a = make_args(dataset, Tp, Tf)
    a.nz, a.qnet_mlp, a.share_eps, a.train_w_mean = 32, [512, 256], True, True   # trainsampler.py:59-62
    a.kld_weight, a.kld_min_clamp, a.recon_weight = 0.1, 10.0, 5.0               # trainsampler.py:90-92
    return a


SAMPLER_CASES = (('eth', 'eth', 8, 12, ('mean', 'shared', 'peragent')), ('nba', 'nba', 5, 10, ('mean',)),
                 ('nba1', 'nba', 5, 10, ('shared', 'peragent')))


def sampler_case_inputs(g, tag, dataset):
    """Inputs of one tests/golden/sampler.npz case: (set_data kwargs for the model, future [n,Tf,2])."""
    from sttode_amd import scenes
    if dataset == 'eth':
        return dict(obs=g['eth_obs'], pred=g['eth_pred']), np.ascontiguousarray(g['eth_pred'].transpose(0, 2, 1))
    d = scenes.nba_batch(int(g[tag + '_seed']), int(g[tag + '_B']))
    return dict(data=d), d['future_traj'].reshape(-1, d['future_traj'].shape[2], 2)


def oracle_sampler(dataset='eth', Tp=8, Tf=12):
    from oracle.sampler_ref import SamplerRef
    from sttode_amd.weights import make_sampler_weights, to_torch_state_dict
    s = SamplerRef(sampler_args(dataset, Tp, Tf)).eval()
    s.load_state_dict(to_torch_state_dict(make_sampler_weights()), strict=True)
    return s


def yardstick_close(got, ref, f64, rtol, atol, what='', factor=4.0):
    """Host-independent comparison of two correct fp32 evaluations (``got``: the oracle on THIS host, ``ref``: the reference's fixture made on
    the authoring container) of an ill-conditioned quantity: each may sit as far from the rounding-free value as fp32 summation order
    puts it, so the band is  atol + rtol |ref| + factor * max|ref - f64|  (``f64``: the same graph evaluated in float64 here)."""
    got, ref, f64 = (np.asarray(v, np.float64) for v in (got, ref, f64))
    assert got.shape == ref.shape == f64.shape, (what, got.shape, ref.shape, f64.shape)
    fin = np.isfinite(ref)
    assert (np.isfinite(got) == fin).all(), what
    spread = np.abs(ref - f64)[fin].max() if fin.any() else 0.0
    err = np.abs(got - ref)[fin]
    bound = atol + rtol * np.abs(ref)[fin] + factor * spread
    assert (err <= bound).all(), f'{what}: max err {err.max():.3e}, fixture-vs-float64 spread {spread:.3e}'


def oracle_sampler_case(g, tag, dataset, Tp, Tf, mode, grads=False, double=False):
    """Runs one sampler.npz case on the oracle: returns (dec, mu, logvar, pred_traj, [total, kld, diverse]) and, with
    ``grads``, additionally name -> gradient of the Sampler's parameters (trainsampler.py:148-150).
    ``double``: the same graph in float64 (the rounding-free yardstick for two fp32 evaluations)."""
    from oracle import sampler_ref as SR
    net, smp = oracle_model(dataset, Tp, Tf), oracle_sampler(dataset, Tp, Tf)
    if double:                                       # fresh float64 copies (the cached fp32 oracle keeps graph tensors as attributes)
        from oracle.sttode_ref import STTODENetRef
        net64 = STTODENetRef(make_args(dataset, Tp, Tf)).eval()
        net64.load_state_dict(net.state_dict(), strict=True)
        net, smp = net64.double(), smp.double()
    inp, fut = sampler_case_inputs(g, tag, dataset)
    smp.share_eps = mode != 'peragent'
    smp.zero_grad()
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64 if double else torch.float32)
    cast = (lambda t: t.double()) if double else (lambda t: t)
    try:
      with torch.set_grad_enabled(grads):
        if dataset == 'eth':
            net.set_data(None, torch.from_numpy(inp['obs']), torch.from_numpy(inp['pred']))
        else:
            net.set_data_nba({k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in inp['data'].items()})
        if double:
            for attr in ('inputs', 'inputs_for_posterior', 'past_traj', 'future_traj', 'cur_location', 'scene_orig'):
                if isinstance(getattr(net, attr, None), torch.Tensor):
                    setattr(net, attr, getattr(net, attr).double())
        dec, sd, vd, aw = smp.forward(net, mean=(mode == 'mean'), eps=cast(torch.from_numpy(g[f'{tag}_{mode}_eps'])))
        tot, ld = SR.compute_sampler_loss(smp.args, cast(torch.from_numpy(fut)), dec.reshape(-1, 20, Tf, 2), vd, sd, {'weight': 1, 'scale': 1.0})
        if grads:
            tot.backward()
    finally:
        torch.set_default_dtype(prev)
    res = (dec.detach().numpy(), sd.mu.detach().numpy(), sd.logvar.detach().numpy(), aw.detach().numpy(),
           np.array([float(tot.detach()), float(ld['kld'].detach()), float(ld['diverse'].detach())]))
    if grads:
        gr = {k: (p.grad.clone() if p.grad is not None else None) for k, p in smp.named_parameters()}
        smp.zero_grad()
        net.zero_grad()
        return res + (gr,)
    return res


def grad_digest(t):
    """Same digest as tests/golden/make_golden.py:grad_summary."""
    g = t.detach().double().flatten().cpu()
    return np.concatenate([[g.sum().item(), g.norm().item(), g.abs().max().item()], g[:48].numpy()])


def grad_case_setup(g, tag, model, dev='cpu'):
    """Feeds one forward_grads.npz case into ``model`` (oracle or HIP); returns the three injected noises as tensors."""
    from sttode_amd import scenes
    if tag == 'eth':
        o, p = g['eth_obs'], g['eth_pred']
        n = o.shape[0]
        model.set_data(None, torch.from_numpy(o), torch.from_numpy(p), torch.ones(n, o.shape[2]), torch.ones(n, p.shape[2]))
    else:
        d = scenes.nba_batch(int(g['nba_seed']), int(g['nba_B']))
        model.set_data_nba({k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in d.items()})
    return tuple(torch.from_numpy(g[f'{tag}_eps_{k}']).to(dev) for k in ('q', 'p1', 'p20'))


def oracle_grads(tag, dataset, Tp, Tf, g, drop=None, double=False):
    """Autograd gradients of the oracle objective: name -> tensor (None for unused parameters), and the 5 loss values.
    ``double``: evaluate the same graph in float64 (the rounding-free yardstick for two fp32 implementations)."""
    m = oracle_model(dataset, Tp, Tf)
    if double:
        from oracle.sttode_ref import STTODENetRef
        m64 = STTODENetRef(make_args(dataset, Tp, Tf)).eval()
        m64.load_state_dict(m.state_dict(), strict=True)
        m = m64.double()
    m.zero_grad()
    eq, ep1, ep20 = grad_case_setup(g, tag, m)
    if double:
        for attr in ('inputs', 'inputs_for_posterior', 'past_traj', 'future_traj', 'cur_location', 'scene_orig'):
            setattr(m, attr, getattr(m, attr).double())
        eq, ep1, ep20 = eq.double(), ep1.double(), ep20.double()
        drop = tuple(d.double() for d in drop) if drop is not None else None
    m.past_encoder.pos_encoder.drop_mask, m.future_encoder.pos_encoder.drop_mask = drop if drop is not None else (None, None)
    prev = torch.get_default_dtype()
    try:
        torch.set_default_dtype(torch.float64 if double else torch.float32)
        vals = m.forward_loss_tensors(eq, ep1, ep20)
        vals[0].backward()
    finally:
        torch.set_default_dtype(prev)
        m.past_encoder.pos_encoder.drop_mask = m.future_encoder.pos_encoder.drop_mask = None
    grads = {k: (p.grad.clone() if p.grad is not None else None) for k, p in m.named_parameters()}
    m.zero_grad()
    return grads, [float(v.detach()) for v in vals]


def horizon_metrics_np(pred_nk, gt, scale=1.0):
    """NumPy restatement of csrc/frontend.hip horizon_metrics_kernel, fp32 in the kernel's order: d = |scale (pred - gt)| (bok_dist: the two
    squared differences added, then sqrt), a running sum over the frames, sum_h / h, minimum over the K samples.  pred_nk [n,K,Tf,2],
    gt [n,Tf,2] -> [n,Tf,2]."""
    s = np.float32(scale)
    dx = (pred_nk[..., 0] - gt[:, None, :, 0]).astype(np.float32) * s
    dy = (pred_nk[..., 1] - gt[:, None, :, 1]).astype(np.float32) * s
    # sqrtf(fmaf(dx, dx, dy * dy)): the fused multiply-add through float64 (dx^2 is exact there; the sum rounds once more only when the two
    # terms' exponents differ by more than 5 bits AND the float64 result sits on a float32 tie: never seen, ~2^-29 per element)
    s2 = dx.astype(np.float64) * dx.astype(np.float64) + (dy * dy).astype(np.float32).astype(np.float64)
    d = np.sqrt(s2.astype(np.float32)).astype(np.float32)                           # [n, K, Tf]
    cum = np.zeros_like(d)
    run = np.zeros(d.shape[:2], np.float32)
    for t in range(d.shape[2]):
        run = (run + d[:, :, t]).astype(np.float32)
        cum[:, :, t] = run / np.float32(t + 1)
    return np.stack([cum.min(axis=1), d.min(axis=1)], axis=-1)


# ---------------------------------------------------------------------------------------------------------------------------------
# Non-default hyper-parameters (round 5): the reference's CLI accepts --hidden_dim / --zdim / --num_decompose / --past_length /
# --future_length (train.py:25-26,37-40).  The cases of tests/golden/dims.npz (made by tests/golden/make_dims_golden.py from the imported
# reference): inputs, latents and noises are REGENERATED from seeds here, the fixture stores the reference's outputs only.
# ---------------------------------------------------------------------------------------------------------------------------------
DIMS_CASES = {
    'tf28': dict(Tf_eth=28, Tf_nba=28),          # the future_length 25-32 hole of rounds 1-4 (16-wide output tiles: NOY = 4)
    'tf36_tp8': dict(Tf_eth=36, Tf_nba=36),      # NOY = 5 with one 16-wide input tile: a (TPX, NOY) pair rounds 1-4 did not build
    'tp20': dict(Tp_eth=20, Tp_nba=20),          # past_length > 16
    'tf60': dict(Tf_eth=60, Tf_nba=60),          # future_length > 48
    'nd1': dict(num_decompose=1),
    'nd3': dict(num_decompose=3),
    'zd16': dict(zdim=16),
    'zd64': dict(zdim=64),
    'hd32': dict(hidden_dim=32),
    'hd128': dict(hidden_dim=128),
    'mix': dict(hidden_dim=32, zdim=16, num_decompose=3, Tp_eth=6, Tf_eth=9, Tp_nba=4, Tf_nba=7),
}
DIMS_GRAD_CASES = ('nd3', 'zd16', 'hd32', 'hd128', 'tf28')   # forward() losses + backward() digests as well


def dims_case_args(tag, dataset):
    c = DIMS_CASES[tag]
    ds = 'nba' if dataset == 'nba' else 'eth'
    a = make_args(dataset, c.get('Tp_' + ds, 5 if ds == 'nba' else 8), c.get('Tf_' + ds, 10 if ds == 'nba' else 12))
    a.hidden_dim, a.zdim, a.num_decompose = c.get('hidden_dim', 64), c.get('zdim', 32), c.get('num_decompose', 2)
    return a


def dims_case_weights(a, seed=1234):
    from sttode_amd.weights import make_weights
    return make_weights(seed, past_length=a.past_length, future_length=a.future_length, hidden_dim=a.hidden_dim, zdim=a.zdim,
                        num_decompose=a.num_decompose)


def dims_case_inputs(tag, dataset):
    """-> (args, inputs, z, (eps_q, eps_p1, eps_p20)): inputs = (obs [N,2,Tp], pred [N,2,Tf]) for 'eth', the loader dict for 'nba'."""
    from sttode_amd import scenes
    ci = list(DIMS_CASES).index(tag)
    a = dims_case_args(tag, dataset)
    Tp, Tf, zd = a.past_length, a.future_length, a.zdim
    rng = np.random.default_rng(4000 + 10 * ci + (dataset == 'nba'))
    if dataset == 'eth':
        inputs = scenes.eth_scene(7300 + ci, n_min=6, n_max=6, obs_len=Tp, pred_len=Tf)
        n = 6
    else:
        # (seed base 1640: no ReLU pre-activation of the encoders' FFN within 6e-6 of zero in any gradient case -- with base 640 one sat at 2.4e-7,
        # where two correct fp32 forwards disagree about the unit's mask and the gradient, a step function there, differs by 5 %)
        inputs = scenes.nba_batch(1640 + ci, 3, N=11, obs_len=Tp, pred_len=Tf)
        n = 33
    z = rng.standard_normal((n * 20, zd)).astype(np.float32)
    eps = tuple(rng.standard_normal(sh).astype(np.float32) for sh in ((n, zd), (n, zd), (n * 20, zd)))
    return a, inputs, z, eps
