#!/usr/bin/env python3
"""Golden vectors for the dataset front-ends: run the REFERENCE loaders (utils/dataloader.py, utils/sddloader.py,
data/dataloader_nba.py) on small synthetic files and store the files + the reference's outputs.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_loader_golden.py        (authoring container only)
"""
import os
import pickle
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.environ.get('STTODE_REFERENCE', '/root/reference'))


def synth_csv(rng, n_frames=40, n_peds=9):
    rows = []
    for p in range(n_peds):
        t0 = int(rng.integers(0, 12))
        t1 = int(rng.integers(t0 + 6, n_frames + 1))
        start, vel = rng.uniform(0, 15, 2), rng.normal(0, 0.4, 2)
        for t in range(t0, t1):
            if p == 3 and t == t0 + 5:
                continue  # a pedestrian with a missing frame: never fully observed across that gap
            x, y = start + vel * (t - t0) + rng.normal(0, 0.02, 2)
            rows.append((10 * t, p + 1, x, y))
    rows.sort()
    return np.asarray(rows).T  # 4 rows: frame, ped, x, y


def main():
    from utils.dataloader import TrajectoryDataset
    from utils.sddloader import SDD_Dataset
    rng = np.random.default_rng(11)
    csv = synth_csv(rng)
    out = {'csv': csv}
    with tempfile.TemporaryDirectory() as d:
        np.savetxt(os.path.join(d, 'synth.csv'), csv, delimiter=',', fmt='%.6f')
        for tag, kw in (('a', dict(obs_len=8, pred_len=12, skip=1, min_ped=1)), ('b', dict(obs_len=5, pred_len=7, skip=2, min_ped=0, traj_scale=2.0))):
            ds = TrajectoryDataset(d, **kw)
            out.update({f'{tag}_obs': ds.obs_traj.numpy(), f'{tag}_pred': ds.pred_traj.numpy(), f'{tag}_obs_rel': ds.obs_traj_rel.numpy(),
                        f'{tag}_ptr': np.asarray([0] + [e for _, e in ds.seq_start_end]), f'{tag}_frame': ds.frame_idx.numpy(),
                        f'{tag}_nonlin': ds.non_linear_ped.numpy(), f'{tag}_valid': ds.valid_ped.numpy()})
    groups = [rng.uniform(0, 1400, (n, 20, 2)) for n in (1, 4, 2, 7)]
    with tempfile.TemporaryDirectory() as d:
        with open(os.path.join(d, 'sdd.pkl'), 'wb') as f:
            pickle.dump(groups, f)
        ds = SDD_Dataset(d, obs_len=8, pred_len=12, traj_scale=50.0)
        out.update(sdd_groups=np.concatenate(groups), sdd_counts=np.asarray([g.shape[0] for g in groups]), sdd_obs=ds.obs_traj.numpy(),
                   sdd_pred=ds.pred_traj.numpy(), sdd_rel=ds.pred_traj_rel.numpy(), sdd_ptr=np.asarray([0] + [e for _, e in ds.seq_start_end]))
    np.savez(os.path.join(HERE, 'loaders.npz'), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == '__main__':
    main()
