#!/usr/bin/env python3
"""Golden vectors for the NBA evaluation metric: the REFERENCE's own loop ``test.test_model_all`` (test.py:495-587: per DataLoader batch
``set_data_nba`` -> ``inference`` -> per-horizon min-over-K mean / final displacement, weighted by the batch size) is run on canned
predictions -- a stub model whose ``inference`` returns stored tensors, a list as the loader -- and the eight figures it prints are stored
with the inputs.  Data only: predictions, ground truth, the printed values.

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/make_nba_eval_golden.py        (authoring container only)
"""
import contextlib
import io
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)


class _Stub:
    """What test_model_all touches of a model: set_data_nba(data) and inference(data) -> [K, B*N, Tf, 2]."""

    def __init__(self, preds):
        self.preds, self.i = preds, 0

    def set_data_nba(self, data):
        pass

    def inference(self, data):
        p = self.preds[self.i]
        self.i += 1
        return torch.from_numpy(p)


def main():
    from make_golden import install_shims
    install_shims()
    sys.argv = [sys.argv[0]]
    cwd = os.getcwd()
    os.chdir(os.environ.get('STTODE_REFERENCE', '/root/reference'))    # test.py does sys.path.append(os.getcwd())
    try:
        import test as ref_test                                        # the reference's test.py
    finally:
        os.chdir(cwd)
    rng = np.random.default_rng(23)
    N, Tf, K = 11, 10, 20
    sizes = [6, 6, 4]                                                  # three loader batches, the last one smaller (test.py:616-622: drop_last is off)
    futs = [rng.uniform(0, 28, (B, N, Tf, 2)).astype(np.float32) for B in sizes]
    preds = [(f.reshape(B * N, Tf, 2)[None] + rng.standard_normal((K, B * N, Tf, 2)).astype(np.float32) * 1.5).astype(np.float32)
             for f, B in zip(futs, sizes)]
    out = {}
    for scale in (1, 3):
        args = types.SimpleNamespace(traj_scale=scale, future_length=Tf)
        loader = [{'future_traj': torch.from_numpy(f), 'past_traj': torch.zeros(f.shape[0], N, 5, 2)} for f in futs]
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            ref_test.test_model_all(_Stub(preds), loader, args)
        vals = {}
        for line in buf.getvalue().splitlines():
            if line.startswith(('ADE', 'FDE')):
                k, v = line.split(':')
                vals[k.strip()] = float(v)
        assert len(vals) == 8, buf.getvalue()
        out[f'scale{scale}_printed'] = np.array([vals[f'{m} {s}.0s'] for m in ('ADE', 'FDE') for s in (1, 2, 3, 4)], np.float64)
    for i, (f, p) in enumerate(zip(futs, preds)):
        out[f'fut{i}'], out[f'pred{i}'] = f, p
    out['n_batches'] = np.int64(len(sizes))
    np.savez(os.path.join(HERE, 'nba_eval.npz'), **out)
    print({k: (v if v.size <= 8 else v.shape) for k, v in out.items()})


if __name__ == '__main__':
    main()
