"""Every native call of ONE NBA-size training step (32 scenes x 11 agents, train.py:59-71), replayed alone: microseconds per call and,
for the GEMM entry points, the fraction of the fp32-MFMA peak (157.3 TFLOP/s).  The calls are recorded from an eager step (tensors kept
alive), then each distinct (entry point, shape) is timed over 30 back-to-back launches with HIP events."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(sys.path[0], 'tests'))
from helpers import make_args
from sttode_amd import STTODENet, scenes, capi
from sttode_amd.weights import make_weights, to_torch_state_dict
dev = torch.device('cuda')
ETH = len(sys.argv) > 1 and sys.argv[1] == 'eth'          # `eth`: ONE scene of 10 pedestrians per step (train.py:72-95) instead of the NBA batch
Tp, Tf = (8, 12) if ETH else (5, 10)
m = STTODENet(make_args('eth' if ETH else 'nba', Tp, Tf), dev); m.load_state_dict(to_torch_state_dict(make_weights(1234, past_length=Tp, future_length=Tf))); m.train()
m.train_graphs = False
if ETH:
    ob, pr = scenes.eth_scene(1, n_min=10, n_max=10)
    def step():
        m.set_data(None, torch.from_numpy(ob), torch.from_numpy(pr), None, None); tot = m.forward()[0]; tot.backward()
else:
    d = scenes.nba_batch(1, 32)
    data = {k: (torch.from_numpy(v) if hasattr(v, 'shape') else v) for k, v in d.items()}
    def step():
        m.set_data_nba(data); tot = m.forward()[0]; tot.backward()
for _ in range(2): step()
rec = []
orig = capi.call
def spy(name, *args, tag=None):
    rec.append((name, args)); return orig(name, *args, tag=tag)
capi.call = spy
import sttode_amd.training as T
step(); torch.cuda.synchronize()
capi.call = orig
def shape_of(name, a):
    if name == 'sttode_tlinear': return ('cols %d J %d I %d trans %d xdiv %d' % (a[11], a[12], a[13], a[5], a[2]), 2.0 * a[11] * a[12] * a[13])
    if name == 'sttode_twgrad': return ('cols %d N %d K %d xdiv %d' % (a[8], a[9], a[10], a[4]), 2.0 * a[8] * a[9] * a[10])
    if name == 'sttode_tlinear_bwd': return ('cols %d N %d K %d Kdx %d' % (a[16], a[17], a[18], a[8]), 2.0 * a[16] * a[17] * (a[18] + a[8]))
    return (' '.join(str(x) for x in a if isinstance(x, int) and not isinstance(x, bool) and 0 < x < 1 << 24)[:60], 0.0)
groups = {}
for name, a in rec:
    s, fl = shape_of(name, a)
    groups.setdefault((name, s), [0, a, fl])[0] += 1
rows = []
for (name, s), (cnt, a, fl) in groups.items():
    for _ in range(3): orig(name, *a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(30): orig(name, *a)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 30 * 1e3
    rows.append((cnt * us, name, s, cnt, us, fl / (us * 1e-6) / 157.3e12 if fl else None))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f'{len(rec)} native calls per step, {tot:.0f} us when each runs alone back to back')
for t, name, s, cnt, us, fr in rows[:(80 if ETH else 45)]:
    print(f'{name[7:]:22s} {s:46s} x{cnt:3d} {us:7.1f} us  sum {t:7.0f}' + (f'  {fr:.2f} of peak' if fr else ''))
