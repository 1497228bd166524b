"""Dataset front-ends that produce the loader tuples of the reference AND the flattened CSR scene batch
(the callers / data formats on the input side of the hot path).

On-disk formats (reference):
  * ETH/UCY  ``<dir>/*.csv`` with four ROWS ``frame, ped, x, y`` (utils/dataloader.py:27-29); sliding windows of
    obs_len + pred_len consecutive frames, pedestrians kept when present in every frame of the window, scenes kept when
    more than ``min_ped`` pedestrians remain (utils/dataloader.py:77-136); coordinates rounded to 4 decimals (:112).
  * SDD      a pickle holding a list of ``[N_i, T, 2]`` arrays (utils/sddloader.py:47-58), divided by ``traj_scale``.
  * NBA      ``.npy [S, T, 11, 2]`` in feet, scaled by 28/94 to metres (data/dataloader_nba.py:35-50).
``__getitem__`` returns the reference's 10-tuple (utils/dataloader.py:186-196) so test.py-style loops keep working;
``scene_batch`` returns many scenes at once as a :class:`sttode_amd.scenes.SceneBatch` for the batched HIP path.
"""
import math
import os
import pickle

import numpy as np
import torch

from .scenes import SceneBatch


def _poly_fit_nonlinear(traj, traj_len, threshold):
    """1.0 if a 2nd-order fit of the last traj_len points leaves residual >= threshold (utils/dataloader.py:9-24)."""
    t = np.linspace(0, traj_len - 1, traj_len)
    rx = np.polyfit(t, traj[0, -traj_len:], 2, full=True)[1]
    ry = np.polyfit(t, traj[1, -traj_len:], 2, full=True)[1]
    return 1.0 if rx + ry >= threshold else 0.0


class _SceneDataset(torch.utils.data.Dataset):
    """Common storage: pedestrian-major arrays + seq_start_end (== CSR scene_ptr)."""

    def _finish(self, seq, seq_rel, loss_mask, non_linear, valid, frame_idx, seq_name, counts):
        o = self.obs_len
        f = lambda a: torch.from_numpy(np.ascontiguousarray(a)).type(torch.float)
        self.obs_traj, self.pred_traj = f(seq[:, :, :o]), f(seq[:, :, o:])
        self.obs_traj_rel, self.pred_traj_rel = f(seq_rel[:, :, :o]), f(seq_rel[:, :, o:])
        self.obs_loss_mask, self.pred_loss_mask = f(loss_mask[:, :o]), f(loss_mask[:, o:])
        self.non_linear_ped, self.valid_ped, self.frame_idx = f(non_linear), f(valid), f(frame_idx)
        self.seq_name = seq_name
        ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        self.scene_ptr = ptr
        self.seq_start_end = [(int(a), int(b)) for a, b in zip(ptr[:-1], ptr[1:])]
        self.num_seq = len(counts)

    def __len__(self):
        return self.num_seq

    def __getitem__(self, index):
        s, e = self.seq_start_end[index]
        return [self.obs_traj[s:e], self.pred_traj[s:e], self.obs_traj_rel[s:e], self.pred_traj_rel[s:e], self.non_linear_ped[s:e],
                self.valid_ped[s:e], self.obs_loss_mask[s:e], self.pred_loss_mask[s:e], self.frame_idx[index], self.seq_name[index]]

    def scene_batch(self, indices=None):
        """Scenes ``indices`` (default: all) as one CSR batch: past [n,Tp,2], future [n,Tf,2], scene_ptr.  No Python loop over the scenes:
        the agents of a scene are consecutive rows of obs_traj / pred_traj, so a batch is one row gather (a plain slice when the scenes are
        consecutive too) -- at 512 scenes per call the per-scene loop took about as long as the GPU needs for the batch."""
        se = getattr(self, '_se_np', None)
        if se is None or len(se) != self.num_seq:
            se = self._se_np = np.asarray(self.seq_start_end, dtype=np.int64).reshape(-1, 2)
        idx = np.arange(self.num_seq) if indices is None else np.asarray(indices if hasattr(indices, '__array__') else list(indices), dtype=np.int64)
        if idx.size == 0:
            raise ValueError('scene_batch: no scenes')
        s, e = se[idx, 0], se[idx, 1]
        cnt = e - s
        ptr = np.zeros(idx.size + 1, np.int64)
        np.cumsum(cnt, out=ptr[1:])
        if idx.size == 1 or bool((s[1:] == e[:-1]).all()):
            rows = slice(int(s[0]), int(e[-1]))                  # consecutive scenes: one contiguous block of agents
        else:
            rows = np.repeat(s - ptr[:-1], cnt) + np.arange(ptr[-1])
        # (NumPy views of the tensors' storage: one copy per array; small torch CPU ops pay a thread-pool hand-off each)
        past = np.ascontiguousarray(self.obs_traj.numpy()[rows].transpose(0, 2, 1))
        fut = np.ascontiguousarray(self.pred_traj.numpy()[rows].transpose(0, 2, 1))
        return SceneBatch(past, fut, ptr.astype(np.int32))


class TrajectoryDataset(_SceneDataset):
    """ETH/UCY CSV windows; same arguments and outputs as the reference class of the same name."""

    def __init__(self, data_dir, obs_len=8, pred_len=8, skip=1, threshold=0.002, min_ped=1, delim='\t', traj_scale=1.0, files=None):
        self.data_dir, self.obs_len, self.pred_len, self.skip = data_dir, obs_len, pred_len, skip
        self.seq_len = L = obs_len + pred_len
        self.max_peds_in_frame = 0
        seqs, rels, masks, nonlin, valid, frame_id, names, counts = [], [], [], [], [], [], [], []
        for fname in (files if files is not None else os.listdir(data_dir)):
            data = np.loadtxt(os.path.join(data_dir, fname), delimiter=',').transpose()  # rows: frame, ped, x, y
            frames = np.unique(data[:, 0])
            fpos = {fr: i for i, fr in enumerate(frames.tolist())}
            order = np.argsort(data[:, 0], kind='stable')
            data = data[order]
            starts = np.searchsorted(data[:, 0], frames, side='left')
            ends = np.append(starts[1:], len(data))
            nseq = int(math.ceil((len(frames) - L + 1) / skip))
            for idx in range(0, nseq * skip + 1, skip):
                hi = min(idx + L, len(frames))
                if idx >= hi:
                    continue
                win = data[starts[idx]:ends[hi - 1]]
                peds = np.unique(win[:, 1])
                self.max_peds_in_frame = max(self.max_peds_in_frame, len(peds))
                cs, cr, ids, nl = [], [], [], []
                for pid in peds:
                    rows = np.around(win[win[:, 1] == pid], decimals=4)
                    front = fpos[rows[0, 0]] - idx
                    end = fpos[rows[-1, 0]] - idx + 1
                    if end - front != L or len(rows) != L:
                        continue  # not present in every frame of the window (the reference would also need len(rows) == L)
                    xy = rows[:, 2:].T / traj_scale
                    rel = np.zeros_like(xy)
                    rel[:, 1:] = xy[:, 1:] - xy[:, :-1]
                    cs.append(xy); cr.append(rel); ids.append(pid); nl.append(_poly_fit_nonlinear(xy, pred_len, threshold))
                if len(cs) > min_ped:
                    seqs.append(np.stack(cs)); rels.append(np.stack(cr)); masks.append(np.ones((len(cs), L)))
                    nonlin += nl; valid += ids; counts.append(len(cs))
                    frame_id.append(frames[idx + obs_len]); names.append(fname)
        if not counts:
            raise ValueError(f'no scene with more than {min_ped} fully observed pedestrians in {data_dir}')
        self._finish(np.concatenate(seqs), np.concatenate(rels), np.concatenate(masks), np.asarray(nonlin), np.asarray(valid),
                     np.asarray(frame_id), names, counts)


class SDD_Dataset(_SceneDataset):
    """Stanford Drone pickle (list of [N_i, T, 2]); arguments as the reference (utils/sddloader.py)."""

    def __init__(self, data_dir, obs_len=8, pred_len=8, skip=1, threshold=0.002, min_ped=1, delim='\t', traj_scale=1.0, file=None):
        self.data_dir, self.obs_len, self.pred_len, self.traj_scale = data_dir, obs_len, pred_len, traj_scale
        self.seq_len = obs_len + pred_len
        fname = file if file is not None else os.listdir(data_dir)[0]
        with open(os.path.join(data_dir, fname), 'rb') as f:
            groups = pickle.load(f)
        counts = [g.shape[0] for g in groups]
        seq = (np.concatenate(groups, axis=0) / traj_scale).transpose(0, 2, 1)
        rel = np.zeros(seq.shape)
        rel[:, :, 1:] = seq[:, :, 1:] - seq[:, :, :-1]
        n = int(np.sum(counts))
        self._finish(seq, rel, np.ones((n, seq.shape[2])), np.ones(n), np.ones(n), np.arange(1, len(counts) + 1), ['sdd'] * len(counts), counts)


class NBADataset(torch.utils.data.Dataset):
    """NBA .npy [S, T, 11, 2] in feet -> metres; items are (past [N,Tp,2], future [N,Tf,2]) (data/dataloader_nba.py:20-61)."""

    def __init__(self, obs_len=5, pred_len=10, training=True, data_root=None):
        self.obs_len, self.pred_len, self.seq_len = obs_len, pred_len, obs_len + pred_len
        if data_root is None:
            data_root = 'datasets/nba/train.npy' if training else 'datasets/nba/test.npy'
        trajs = np.load(data_root) / (94 / 28)
        trajs = trajs[:32500] if training else trajs[:12500]
        self.batch_len = len(trajs)
        self.traj_abs = torch.from_numpy(trajs).type(torch.float).permute(0, 2, 1, 3)  # [S, N, T, 2]

    def __len__(self):
        return self.batch_len

    def __getitem__(self, index):
        return [self.traj_abs[index, :, :self.obs_len, :], self.traj_abs[index, :, self.obs_len:, :]]


def seq_collate(data):
    """data/dataloader_nba.py:7-18."""
    past, fut = zip(*data)
    return {'past_traj': torch.stack(past, dim=0), 'future_traj': torch.stack(fut, dim=0), 'seq': 'nba'}
