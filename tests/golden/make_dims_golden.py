#!/usr/bin/env python3
"""Golden vectors for NON-DEFAULT hyper-parameters: the reference's CLI accepts --hidden_dim / --zdim / --num_decompose / --past_length /
--future_length (train.py:25-26,37-40) and every dimension derives from them (model/STTODE.py:182-196,242-260,309-318,359-361).  One ETH scene
and one NBA batch per new value through the IMPORTED reference: ``inference()`` with injected latents, and for a subset the training
objective ``forward()`` + ``backward()`` digests.  Data only (inputs, latents, outputs); weights come from the NumPy recipe
``make_weights(seed, **hyper-parameters)``.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_dims_golden.py        (authoring container only)
"""
import argparse
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)

def main():
    from make_golden import NoiseQueue, capture, grad_summary, install_shims, npy
    install_shims()
    from model.STTODE import STTODENet
    from sttode_amd import scenes
    from sttode_amd.weights import to_torch_state_dict
    noise = NoiseQueue()
    out = {}
    sys.path.insert(0, os.path.join(os.path.dirname(HERE)))          # tests/helpers.py: the case table, shared with the tests
    from helpers import DIMS_CASES, DIMS_GRAD_CASES, dims_case_inputs, dims_case_weights
    for tag in DIMS_CASES:
        for dataset in ('eth', 'nba'):
            a, inputs, z, (e_q, e_p, e20) = dims_case_inputs(tag, dataset)
            Tp, Tf = a.past_length, a.future_length
            m = STTODENet(a, torch.device('cpu')).eval()
            m.load_state_dict(to_torch_state_dict(dims_case_weights(a)), strict=True)
            k = f'{tag}_{dataset}_'
            if dataset == 'eth':
                o, p = inputs
                n = o.shape[0]
                setd = lambda: m.set_data(None, torch.from_numpy(o), torch.from_numpy(p), torch.ones(n, Tp), torch.ones(n, Tf))
                data = None
            else:
                data = {kk: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for kk, v in inputs.items()}
                setd = lambda: m.set_data_nba(data)
            cap = {}
            hk = capture(m.past_encoder, cap, 'pf')
            with torch.no_grad():
                setd()
                noise.push(z)
                pred = m.inference(data)                                  # [K, n, Tf, 2]
            hk.remove()
            out[k + 'out'], out[k + 'past_feature'] = npy(pred), npy(cap['pf'])
            if tag in DIMS_GRAD_CASES:
                m.zero_grad()
                setd()
                noise.push(e_q, e_p, e20)
                vals = m()                                                # forward(): (total, loss_pred, loss_recover, loss_kl, loss_diverse)
                vals[0].backward()
                out[k + 'losses'] = np.array([float(vals[0].detach())] + [float(v) for v in vals[1:]], np.float64)
                for name, prm in m.named_parameters():
                    if prm.grad is None:
                        out[f'{k}nograd::{name}'] = np.int64(1)
                    else:
                        out[f'{k}grad::{name}'] = grad_summary(prm.grad)
    assert not noise.q
    np.savez_compressed(os.path.join(HERE, 'dims.npz'), **out)
    print('dims.npz bytes:', os.path.getsize(os.path.join(HERE, 'dims.npz')), 'cases:', len(DIMS_CASES))


if __name__ == '__main__':
    main()
