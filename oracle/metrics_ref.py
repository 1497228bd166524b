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


def nba_eval_printed(batches, traj_scale=1.0, future_length=10):
    """The eight figures test.py:495-587 prints for an NBA test set: ``batches`` = [(pred_kn [K, B*N, Tf, 2], future [B, N, Tf, 2]), ...] one per
    DataLoader batch.  Per batch and horizon h: mean over agents of min_k (mean displacement over the first h frames / displacement of frame
    h), times the batch size B; summed over batches, divided by the number of scenes (:530-575).  Printed (:577-586, 0.4 s per frame):
    ADE 1.0s = (avg_2 + avg_3) / 2, ADE 2.0s = avg_5, ADE 3.0s = (avg_8 + avg_7) / 2, ADE 4.0s = avg_10; FDE likewise with dest_h.
    Returns float64 [8] = ADE 1..4 s, FDE 1..4 s."""
    assert future_length == 10, 'the reference hard-codes ten horizons'
    avg, dest, all_num = np.zeros(11), np.zeros(11), 0
    for pred_kn, fut in batches:
        B = fut.shape[0]
        y = np.asarray(fut, np.float32).reshape(-1, future_length, 2) * traj_scale
        e = nba_horizon_errors(np.asarray(pred_kn) * traj_scale, y, range(1, 11))
        for h in range(1, 11):
            avg[h] += e[h][0] * B
            dest[h] += e[h][1] * B
        all_num += B
    avg, dest = avg / all_num, dest / all_num
    return np.array([(avg[2] + avg[3]) / 2, avg[5], (avg[8] + avg[7]) / 2, avg[10],
                     (dest[2] + dest[3]) / 2, dest[5], (dest[7] + dest[8]) / 2, dest[10]], np.float64)
