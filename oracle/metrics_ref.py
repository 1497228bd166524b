"""CPU oracle for best-of-K ADE/FDE (TEST INFRASTRUCTURE ONLY).

Restates utils/metrics.py:7-26 (ETH/UCY/SDD) and the NBA horizon variant test.py:530-551
in NumPy.  Pinned by tests/golden (metrics computed by the reference's own functions).
"""
import numpy as np


def best_of_k_ade_fde(pred, gt):
    """pred [n, K, Tf, 2], gt [n, Tf, 2] -> (ade[n], fde[n]) per-agent minima over K."""
    d = np.linalg.norm(pred - gt[:, None], axis=-1)  # [n, K, Tf]
    return d.mean(axis=-1).min(axis=1), d[..., -1].min(axis=1)


def compute_ade(pred, gt):
    """utils/metrics.py:7-15: mean over agents of min_k mean_t ||pred-gt||."""
    return float(best_of_k_ade_fde(pred, gt)[0].mean())


def compute_fde(pred, gt):
    """utils/metrics.py:18-26."""
    return float(best_of_k_ade_fde(pred, gt)[1].mean())


def nba_horizon_errors(pred_kn, gt, horizons):
    """test.py:530-551: pred_kn [K, n, Tf, 2], gt [n, Tf, 2].

    For each horizon h (1-based frame count): avg = mean_n min_k mean_{t<h} dist, dest = mean_n min_k dist[t=h-1].
    """
    d = np.linalg.norm(pred_kn - gt[None], axis=-1)  # [K, n, Tf]
    out = {}
    for h in horizons:
        out[h] = (float(d[:, :, :h].mean(axis=2).min(axis=0).mean()), float(d[:, :, h - 1].min(axis=0).mean()))
    return out
