"""Synthetic scene generators and the flattened (CSR) scene-batch layout.

Layouts mirror the reference loaders so the module is a drop-in for their consumers:
  * ETH/UCY  utils/dataloader.py:186-196 -> per scene ``obs_traj [N,2,Tp]``, ``pred_traj [N,2,Tf]``,
    masks ``[N,Tp]`` / ``[N,Tf]``; scenes concatenated over pedestrians with ``seq_start_end``
    (utils/dataloader.py:177-181), which is the CSR ``scene_ptr`` used here.
  * SDD      utils/sddloader.py:98-109 -> same tuple, ragged N (1..40), pixels / 50.
  * NBA      data/dataloader_nba.py:7-18,35-50 -> dict ``past_traj [B,N,Tp,2]``, ``future_traj [B,N,Tf,2]``.

Generators are NumPy-only and seeded per scene so the GPU box regenerates
bit-identical inputs.  No dataset ships with the reference (``datasets`` is a placeholder).
"""
from dataclasses import dataclass

import numpy as np

SCENE_SEED0 = 20250418


def _walk(rng, n, T, extent, v_sigma, noise, decimals=4):
    start = rng.uniform(0.0, 1.0, size=(n, 1, 2)) * np.asarray(extent, np.float64)
    vel = rng.normal(0.0, v_sigma, size=(n, 1, 2))
    t = np.arange(T, dtype=np.float64)[None, :, None]
    pos = start + vel * t + rng.normal(0.0, noise, size=(n, T, 2))
    return np.around(pos, decimals=decimals)  # utils/dataloader.py:112


def eth_scene(s, obs_len=8, pred_len=12, n_min=2, n_max=32):
    """ETH-shaped scene ``s``: (obs [N,2,Tp], pred [N,2,Tf]) fp32, N ~ U{n_min..n_max}."""
    rng = np.random.default_rng(SCENE_SEED0 + s)
    n = int(rng.integers(n_min, n_max + 1))
    pos = _walk(rng, n, obs_len + pred_len, (15.0, 15.0), 0.4, 0.02).astype(np.float32)
    pos = pos.transpose(0, 2, 1)  # [N, 2, T]
    return np.ascontiguousarray(pos[:, :, :obs_len]), np.ascontiguousarray(pos[:, :, obs_len:])


def ucy_scene(s, obs_len=8, pred_len=12):
    """UCY-mixed: three interleaved streams zara1 / zara2 (2-20 peds) and univ (20-60 peds)."""
    lo, hi = ((2, 20), (2, 20), (20, 60))[s % 3]
    return eth_scene(1_000_000 + s, obs_len, pred_len, lo, hi)


def sdd_scene(s, obs_len=8, pred_len=12, sdd_scale=50.0):
    """SDD-shaped ragged scene: N = 1 + Geometric(0.25) clipped to 40; pixel coordinates / sdd_scale."""
    rng = np.random.default_rng(SCENE_SEED0 + 2_000_000 + s)
    n = int(min(1 + rng.geometric(0.25), 40))
    pos = _walk(rng, n, obs_len + pred_len, (1400.0, 1900.0), 12.0, 1.0, decimals=2) / sdd_scale
    pos = pos.astype(np.float32).transpose(0, 2, 1)
    return np.ascontiguousarray(pos[:, :, :obs_len]), np.ascontiguousarray(pos[:, :, obs_len:])


def nba_batch(seed, B, N=11, obs_len=5, pred_len=10):
    """NBA-shaped batch dict (court 28.65 x 15.24 m, i.e. feet * 28/94, data/dataloader_nba.py:36)."""
    rng = np.random.default_rng(SCENE_SEED0 + 3_000_000 + seed)
    T = obs_len + pred_len
    start = rng.uniform(0.0, 1.0, size=(B, N, 1, 2)) * np.array([28.65, 15.24])
    step = rng.normal(0.0, 0.35, size=(B, N, T, 2))
    pos = (start + np.cumsum(step, axis=2)).astype(np.float32)
    return {'past_traj': np.ascontiguousarray(pos[:, :, :obs_len]),
            'future_traj': np.ascontiguousarray(pos[:, :, obs_len:]), 'seq': 'nba'}


def latents(seed, n_agents, K=20, zdim=32):
    """Injected prior samples z [n*K, zdim], row = agent*K + k (model/STTODE.py:609-616 ordering)."""
    return np.random.default_rng(seed + 1).standard_normal((n_agents * K, zdim)).astype(np.float32)


@dataclass
class SceneBatch:
    """Flattened batch of independent scenes (ETH/UCY/SDD path).

    past   [n, Tp, 2]  world coordinates, agent-major (all pedestrians of scene 0, then scene 1, ...)
    future [n, Tf, 2]  ground truth (metrics only)
    scene_ptr [S+1]    CSR offsets into the agent axis  (== seq_start_end, utils/dataloader.py:177-181)
    """
    past: np.ndarray
    future: np.ndarray
    scene_ptr: np.ndarray

    @property
    def n_agents(self):
        return int(self.scene_ptr[-1])

    @property
    def n_scenes(self):
        return len(self.scene_ptr) - 1

    def scene(self, s):
        a, b = int(self.scene_ptr[s]), int(self.scene_ptr[s + 1])
        # back to the loader layout [N,2,T]
        return (np.ascontiguousarray(self.past[a:b].transpose(0, 2, 1)),
                np.ascontiguousarray(self.future[a:b].transpose(0, 2, 1)))

    def slice_scenes(self, s0, s1):
        a, b = int(self.scene_ptr[s0]), int(self.scene_ptr[s1])
        return SceneBatch(self.past[a:b], self.future[a:b], self.scene_ptr[s0:s1 + 1] - self.scene_ptr[s0])


def make_scene_batch(scene_ids, kind='eth', obs_len=8, pred_len=12):
    gen = {'eth': eth_scene, 'ucy': ucy_scene, 'sdd': sdd_scene}[kind]
    past, fut, ptr = [], [], [0]
    for s in scene_ids:
        o, p = gen(int(s), obs_len, pred_len)
        past.append(o.transpose(0, 2, 1))
        fut.append(p.transpose(0, 2, 1))
        ptr.append(ptr[-1] + o.shape[0])
    return SceneBatch(np.ascontiguousarray(np.concatenate(past, 0)), np.ascontiguousarray(np.concatenate(fut, 0)),
                      np.asarray(ptr, np.int32))


def shard_scenes(scene_ptr, world_size):
    """Contiguous scene ranges balanced by agent count: returns [(s0, s1)] * world_size.
    Every rank gets at least one scene whenever there are at least ``world_size`` scenes (a skewed agent distribution must not
    leave a rank empty: an empty rank would skip the collectives its peers wait in); with fewer scenes than ranks the
    trailing ranks get empty ranges (s0 == s1), which ``parallel.infer_sharded`` turns into zero-row contributions."""
    ptr = np.asarray(scene_ptr, np.int64)
    S, total = len(ptr) - 1, int(ptr[-1])
    cuts = [0]
    for r in range(1, world_size):
        target = total * r / world_size
        c = int(np.searchsorted(ptr, target, side='left'))
        lo = min(cuts[-1] + 1, S)                    # previous rank keeps >= 1 scene (while scenes remain)
        hi = max(S - (world_size - r), lo) if S >= world_size else S   # leave >= 1 scene for every later rank
        cuts.append(int(np.clip(c, lo, hi)))
    cuts.append(S)
    return [(cuts[r], cuts[r + 1]) for r in range(world_size)]
