"""Training-step timing (set_data + forward + backward + Adam) on the HIP training kernels, with a per-entry-point breakdown.
ETH: one scene of 32 agents per step (train.py:72-95); NBA: 32 scenes x 11 agents per step (train.py:59-71, batch_size 32).
(The CPU comparison lives in ``bench.py --train``'s cpu_baseline leg.)  STTODE_TRAIN_GRAPHS=0 disables the hipGraph replay."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(sys.path[0], 'tests'))
from helpers import make_args
from sttode_amd import STTODENet, scenes, capi
from sttode_amd.weights import make_weights, to_torch_state_dict
dev = torch.device('cuda')
for name, ds, Tp, Tf in (('eth N=32', 'eth', 8, 12), ('nba B=32 N=11', 'nba', 5, 10)):
    sd = to_torch_state_dict(make_weights(1234, past_length=Tp, future_length=Tf))
    m = STTODENet(make_args(ds, Tp, Tf), dev); m.load_state_dict(sd); m.train()
    if ds == 'eth':
        ob, pr = scenes.eth_scene(1, n_min=32, n_max=32)
        feed = lambda mod: mod.set_data(None, torch.from_numpy(ob), torch.from_numpy(pr), torch.ones(32, Tp), torch.ones(32, Tf))
        n = 32
    else:
        d = scenes.nba_batch(1, 32)
        data = {k: (torch.from_numpy(v) if hasattr(v, 'shape') else v) for k, v in d.items()}
        feed = lambda mod: mod.set_data_nba(data)
        n = 352
    opt = torch.optim.Adam(m.parameters(), lr=1e-4)
    def step():
        feed(m)
        tot = m.forward()[0]
        opt.zero_grad(); tot.backward(); opt.step()
    for _ in range(3): step()
    torch.cuda.synchronize()
    reps = 20
    t = time.perf_counter()
    for _ in range(reps): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / reps
    # kernel-only time of one step
    m.train_graphs = False
    capi.TIMING = []
    step(); torch.cuda.synchronize()
    kt = sum(e0.elapsed_time(e1) for _, e0, e1 in capi.TIMING); nk = len(capi.TIMING)
    by = {}
    for tag, e0, e1 in capi.TIMING: by[tag] = by.get(tag, 0) + e0.elapsed_time(e1)
    capi.TIMING = None
    print(f'{name}: HIP {dt*1e3:.2f} ms/step wall; eager step: {nk} entry-point calls, {kt:.2f} ms between their HIP events')
    print('   ', {k: round(v, 3) for k, v in sorted(by.items(), key=lambda kv: -kv[1])[:8]})
