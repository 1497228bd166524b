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
