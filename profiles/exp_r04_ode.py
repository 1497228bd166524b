"""BASELINE config 5's "40 RK4 steps" with an attention group > 1 (obs 10 / pred 40, B scenes x 10 agents in ONE attention group): time of
the encoder's integration inside the native NBA call (round 4: one fused launch + one attention per stage = 8 launches per RK4 step).
    gpurun -- python profiles/exp_r04_ode.py [B]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(sys.path[0], 'tests'))
from helpers import make_args
from sttode_amd import STTODENet, scenes
from sttode_amd.weights import make_weights, to_torch_state_dict
dev = torch.device('cuda')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
Tp, Tf, N = 10, 40, 10
m = STTODENet(make_args('nba', Tp, Tf), dev).eval()
m.load_state_dict(to_torch_state_dict(make_weights(1234, past_length=Tp, future_length=Tf)))
d = scenes.nba_batch(1, B, N=N, obs_len=Tp, pred_len=Tf)
data = {'past_traj': torch.from_numpy(d['past_traj']).to(dev), 'future_traj': torch.from_numpy(d['future_traj']).to(dev)}
for method, steps in (('euler', 1), ('rk4', 40), ('rk4_classic', 40), ('euler', 40)):
    m.ode_method, m.ode_steps = method, steps
    m.set_data_nba(data)
    for _ in range(2):
        m.inference(data)
    torch.cuda.synchronize()
    t = time.perf_counter()
    reps = 5
    for _ in range(reps):
        m.inference(data)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / reps
    print(f'B={B} x {N} agents, obs {Tp} / pred {Tf}, {method} x {steps}: {dt * 1e3:.2f} ms per call', flush=True)
