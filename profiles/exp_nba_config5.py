"""Timing of the NBA / long-horizon path at BASELINE config-5 shapes on one GPU (per-stage, via the native timers)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(sys.path[0], 'tests'))
from helpers import make_args
from sttode_amd import STTODENet, scenes
from sttode_amd.weights import make_weights, to_torch_state_dict
dev = torch.device('cuda')
CASES = ((128, 11, 5, 10), (1024, 10, 10, 40), (4096, 10, 10, 40))
if os.environ.get('ONLY_CONFIG5'):
    CASES = CASES[2:]
for (B, N, Tp, Tf) in CASES:
    m = STTODENet(make_args('nba', Tp, Tf), dev).eval()
    m.load_state_dict(to_torch_state_dict(make_weights(1234, past_length=Tp, future_length=Tf)))
    d = scenes.nba_batch(1, B, N=N, obs_len=Tp, pred_len=Tf)
    data = {'past_traj': torch.from_numpy(d['past_traj']).to(dev), 'future_traj': torch.from_numpy(d['future_traj']).to(dev)}
    m.set_data_nba(data)
    for _ in range(2):
        m.inference(data)
    torch.cuda.synchronize()
    m.native().timing(1)
    t = time.perf_counter()
    reps = 5
    for _ in range(reps):
        out = m.inference(data)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / reps
    st = m.native().read_timing()
    print(f'B={B} N={N} Tp={Tp} Tf={Tf}: {dt*1e3:.2f} ms/call, {B*N*20/dt/1e6:.1f} M traj/s', {k: round(v[0] / v[1] * 1e3) for k, v in st.items()})
