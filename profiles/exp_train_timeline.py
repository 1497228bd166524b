"""One-scene training step (train.py:72-95 pattern: set_data, forward, zero_grad, backward, Adam) with the host-side split of a step;
run under `rocprofv3 --kernel-trace` and summarise with profiles/summarize_train_timeline.py for the GPU-side picture."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(sys.path[0], 'tests'))
from helpers import make_args
from sttode_amd import STTODENet, scenes
from sttode_amd.weights import make_weights, to_torch_state_dict
dev = torch.device('cuda')
ds, Tp, Tf = 'eth', 8, 12
m = STTODENet(make_args(ds, Tp, Tf), dev); m.load_state_dict(to_torch_state_dict(make_weights(1234))); m.train()
if os.environ.get('NO_GRAPHS'): m.train_graphs = False
ob, pr = scenes.eth_scene(1, n_min=32, n_max=32)
ob, pr = torch.from_numpy(ob), torch.from_numpy(pr)
mk = (torch.ones(32, Tp), torch.ones(32, Tf))
from sttode_amd.optim import Adam
opt = torch.optim.Adam(m.parameters(), lr=1e-4, fused=True) if os.environ.get('STTODE_TORCH_ADAM') else Adam(m.parameters(), lr=1e-4)
T = [0.0] * 5
def step(rec):
    t0 = time.perf_counter()
    m.set_data(None, ob, pr, *mk)
    t1 = time.perf_counter()
    tot = m.forward()[0]
    t2 = time.perf_counter()
    opt.zero_grad(); tot.backward()
    t3 = time.perf_counter()
    opt.step()
    t4 = time.perf_counter()
    if rec:
        for i, d in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3)): T[i] += d
for _ in range(5): step(False)
torch.cuda.synchronize()
reps = int(os.environ.get('REPS', '50'))
t = time.perf_counter()
for _ in range(reps): step(True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / reps
print(f'eth N=32: {dt * 1e3:.3f} ms/step wall; host split per step (ms): set_data {T[0] / reps * 1e3:.3f}, forward (to the loss values on the host) '
      f'{T[1] / reps * 1e3:.3f}, zero_grad+backward {T[2] / reps * 1e3:.3f}, Adam {T[3] / reps * 1e3:.3f}')
