"""NBA-size training step (32 scenes x 11 agents per step, train.py:59-71) for `rocprofv3 --kernel-trace --stats`."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(sys.path[0], 'tests'))
from helpers import make_args
from sttode_amd import STTODENet, scenes
from sttode_amd.weights import make_weights, to_torch_state_dict
dev = torch.device('cuda')
Tp, Tf = 5, 10
m = STTODENet(make_args('nba', Tp, Tf), dev); m.load_state_dict(to_torch_state_dict(make_weights(1234, past_length=Tp, future_length=Tf))); m.train()
d = scenes.nba_batch(1, 32)
data = {k: (torch.from_numpy(v) if hasattr(v, 'shape') else v) for k, v in d.items()}
from sttode_amd.optim import Adam
opt = torch.optim.Adam(m.parameters(), lr=1e-4, fused=True) if os.environ.get('STTODE_TORCH_ADAM') else Adam(m.parameters(), lr=1e-4)
def step():
    m.set_data_nba(data); tot = m.forward()[0]; opt.zero_grad(); tot.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize(); print(f'nba B=32 N=11: {(time.perf_counter() - t) / 20 * 1e3:.3f} ms/step')
