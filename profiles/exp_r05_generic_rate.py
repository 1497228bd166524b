"""Rate of the GENERIC-DIMENSION form (sttode_amd/generic.py: non-default --hidden_dim / --zdim / --num_decompose / --past_length /
--future_length, train.py:25-26,37-40) against (a) the fused forms on the reference's default widths at the same call size and (b) the CPU
port (oracle/, the checker -- timed here as the baseline only) of the same non-default model on a bounded sample.
One NBA call of 128 scenes x 11 agents x K = 20 per inference(); one scene batch of 64 ETH-like scenes."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(sys.path[0], 'tests'))
from helpers import DIMS_CASES, dims_case_args, dims_case_weights, make_args
from sttode_amd import STTODENet, scenes
from sttode_amd.weights import make_weights, to_torch_state_dict
dev = torch.device('cuda')
K = 20


def gpu_rate(a, weights, B=128, N=11, reps=30):
    m = STTODENet(a, dev).eval()
    m.load_state_dict(to_torch_state_dict(weights))
    d = scenes.nba_batch(777, B, N=N, obs_len=a.past_length, pred_len=a.future_length)
    data = {k: (torch.from_numpy(v).to(dev) if isinstance(v, np.ndarray) else v) for k, v in d.items()}
    z = torch.randn(B * N * K, a.zdim, device=dev)
    m.set_data_nba(data)
    for _ in range(5):
        out = m.inference(data, z=z)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps):
        m.set_data_nba(data)
        out = m.inference(data, z=z)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / reps
    assert bool(torch.isfinite(out).all())
    return dt, B * N * K / dt, m._generic


def cpu_rate(a, weights, B=8, N=11):
    from oracle.sttode_ref import STTODENetRef
    o = STTODENetRef(a).eval()
    o.load_state_dict(to_torch_state_dict(weights), strict=True)
    d = scenes.nba_batch(777, B, N=N, obs_len=a.past_length, pred_len=a.future_length)
    data = {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in d.items()}
    z = torch.randn(B * N * K, a.zdim)
    with torch.no_grad():
        o.set_data_nba(data); o.inference(data, z=z)
        t = time.perf_counter(); reps = 0
        while time.perf_counter() - t < 4.0:
            o.set_data_nba(data); o.inference(data, z=z); reps += 1
    return B * N * K * reps / (time.perf_counter() - t)


a0 = make_args('nba', 5, 10)
dt, r, gen = gpu_rate(a0, make_weights(1234, past_length=5, future_length=10))
print(f'reference defaults (hidden 64, z 32, 2 blocks, obs 5 / pred 10), fused forms (generic={gen}): {dt * 1e3:.3f} ms per call of 128 x 11, {r / 1e6:.2f} M trajectories/s')
for tag in ('hd128', 'hd32', 'zd16', 'zd64', 'nd1', 'nd3', 'tp20', 'tf60', 'mix'):
    a = dims_case_args(tag, 'nba')
    w = dims_case_weights(a)
    dt, r, gen = gpu_rate(a, w)
    c = cpu_rate(a, w)
    print(f'{tag:6s} {DIMS_CASES[tag]} generic={gen}: {dt * 1e3:.3f} ms per call of 128 x 11, {r / 1e6:.2f} M trajectories/s; '
          f'CPU port ({torch.get_num_threads()} threads, 8 x 11 sample): {c / 1e3:.1f} k trajectories/s -> {r / c:.0f}x')
