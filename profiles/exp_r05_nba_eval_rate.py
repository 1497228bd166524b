"""End-to-end NBA evaluation rate (test.py:495-552 flow) over a STORED test set of 64 loader batches of 128 scenes x 11 agents (obs 5 / pred 10,
K = 20): the round-4 loop -- one serial inference() per batch, the per-horizon min-over-K metric as torch ops, two .cpu() syncs per batch -- against
the round-5 flow -- G batches per call (attention within each batch), inference_async, the metric as a HIP kernel on the call's stream."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(sys.path[0], 'tests'))
from helpers import make_args
from sttode_amd import STTODENet, evaluate, scenes
from sttode_amd.weights import make_weights, to_torch_state_dict
dev = torch.device('cuda')
m = STTODENet(make_args('nba', 5, 10), dev).eval()
m.load_state_dict(to_torch_state_dict(make_weights(1234, past_length=5, future_length=10)))
NB, B, N, K = 64, 128, 11, 20
loader = []
for i in range(NB):
    d = scenes.nba_batch(50000 + i, B, N=N)
    loader.append({'past_traj': torch.from_numpy(d['past_traj']).pin_memory(), 'future_traj': torch.from_numpy(d['future_traj']).pin_memory()})
traj = NB * B * N * K
def rate(fn, reps=3):
    fn()
    best = None
    for _ in range(reps):
        torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize()
        dt = time.perf_counter() - t
        best = dt if best is None else min(best, dt)
    return traj / best / 1e6, r
base, r0 = rate(lambda: evaluate.eval_nba(m, loader, pipelined=False))
print(f'serial loop (round 4: inference() per batch + torch metric ops): {base:.1f} M trajectories/s over {NB} batches of {B} x {N}')
for G in (1, 4, 16):
    v, r = rate(lambda: evaluate.eval_nba(m, loader, groups_per_call=G))
    print(f'pipelined, {G:2d} batch(es) per call, horizon metric as a HIP kernel on the call\'s stream: {v:.1f} M trajectories/s ({v / base:.1f}x)')
print('ADE/FDE at 4.0 s (different latents per run; for scale):', tuple(round(x, 3) for x in r[10]), tuple(round(x, 3) for x in r0[10]))
