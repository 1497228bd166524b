"""eval_scenes (test.py:163-208 on top of the batched path) over a stored dataset of 12 288 ETH-shaped scenes, 512 scenes per call:
pipelined calls (default) against one serial inference() + best_of_k per batch."""
import sys, time, torch, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import make_args
from sttode_amd import STTODENet, scenes, datasets
from sttode_amd.evaluate import eval_scenes
from sttode_amd.weights import make_weights, to_torch_state_dict
m = STTODENet(make_args('eth', 8, 12), torch.device('cuda')).eval(); m.load_state_dict(to_torch_state_dict(make_weights(1234)))
class DS(datasets._SceneDataset):
    def __init__(self, n_scenes):
        sb = scenes.make_scene_batch(range(512), 'eth')
        reps = n_scenes // 512
        cnt = np.tile(np.diff(sb.scene_ptr), reps)
        ends = np.cumsum(cnt)
        self.seq_start_end = list(zip((ends - cnt).tolist(), ends.tolist()))
        self.num_seq = len(cnt)
        self.obs_traj = torch.from_numpy(np.ascontiguousarray(np.tile(sb.past.transpose(0, 2, 1), (reps, 1, 1))))
        self.pred_traj = torch.from_numpy(np.ascontiguousarray(np.tile(sb.future.transpose(0, 2, 1), (reps, 1, 1))))
ds = DS(12288)
for pipe in (True, False, True, False):
    torch.manual_seed(0); torch.cuda.synchronize(); t = time.perf_counter()
    a, f, n = eval_scenes(m, ds, pipelined=pipe)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f'eval_scenes over {len(ds)} scenes ({n} agents, {n * 20 / dt / 1e6:.1f} M trajectories/s), pipelined={pipe}: {dt * 1e3:.0f} ms, ADE {a:.4f} FDE {f:.4f}')
