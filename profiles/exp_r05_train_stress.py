"""Free-running training loop stress (round 5: a replayed step hands its loss values to the host from the middle of its graph, so nothing in
train.py:61-67 waits for the end of the queue any more).  1200 steps in train() mode (rotation, dropout, in-graph noise) over 24 ETH scenes of
different sizes -- 24 graphs used in turn, two rotating gradient buffers each, the pinned staging ring reused every fourth step -- and 300
steps over NBA batches of two sizes, (a) free-running and (b) with a device synchronisation after every call, same seeds: the per-step loss
values and the final parameters must be IDENTICAL (replays are deterministic; only the host's position relative to the queue differs)."""
import os, sys, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(sys.path[0], 'tests'))
from helpers import make_args
from sttode_amd import STTODENet, scenes
from sttode_amd.optim import Adam
from sttode_amd.weights import make_weights, to_torch_state_dict
dev = torch.device('cuda')


def loop(ds, steps, sync):
    torch.manual_seed(11); np.random.seed(11)
    if ds == 'eth':
        m = STTODENet(make_args('eth', 8, 12), dev); m.load_state_dict(to_torch_state_dict(make_weights(1234)))
        data = [tuple(torch.from_numpy(x) for x in scenes.eth_scene(200000 + i)) for i in range(24)]
    else:
        m = STTODENet(make_args('nba', 5, 10), dev); m.load_state_dict(to_torch_state_dict(make_weights(1234, past_length=5, future_length=10)))
        data = [{k: (torch.from_numpy(v) if hasattr(v, 'shape') else v) for k, v in scenes.nba_batch(70 + i, 32 if i % 2 else 8).items()} for i in range(6)]
    if os.environ.get('DEVICE_DATA'):                              # the loader's tensors moved to the device by the caller (train.py:77) instead of staged by set_data
        data = [tuple(x.to(dev) for x in d) if ds == 'eth' else {k: (v.to(dev) if hasattr(v, 'shape') else v) for k, v in d.items()} for d in data]
    m.train()
    opt = Adam(m.parameters(), lr=1e-4)
    vals = []
    torch.cuda.synchronize(); t = time.perf_counter()
    ph = [0.0, 0.0, 0.0]
    for i in range(steps):
        d = data[i % len(data)]
        t0 = time.perf_counter()
        if ds == 'eth':
            m.set_data(None, d[0], d[1], None, None)
        else:
            m.set_data_nba(d)
        t1 = time.perf_counter()
        out = m.forward()
        t2 = time.perf_counter()
        if sync: torch.cuda.synchronize()
        opt.zero_grad(); out[0].backward(); opt.step()
        if sync: torch.cuda.synchronize()
        vals.append(out[1:])
        t3 = time.perf_counter()
        ph[0] += t1 - t0; ph[1] += t2 - t1; ph[2] += t3 - t2
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / steps
    print(f'   {ds} sync={sync}: host phases per step (us): set_data {ph[0] / steps * 1e6:.0f}, forward {ph[1] / steps * 1e6:.0f}, zero_grad + backward + step {ph[2] / steps * 1e6:.0f}; '
          f'graphs {len(m._graphs)}; mean agents {sum((x[0].shape[0] if ds == "eth" else x["past_traj"].shape[0] * 11) for x in data) / len(data):.1f}')
    return vals, [p.detach().clone() for p in m.parameters()], dt


for ds, steps in [x for x in (('eth', int(os.environ.get('ETH_STEPS', '1200'))), ('nba', int(os.environ.get('NBA_STEPS', '300')))) if x[1] > 0]:
    loop(ds, 60, False)                                           # (first-launch costs of every kernel: not part of either timing)
    va, pa, ta = loop(ds, steps, False)
    vb, pb, tb = loop(ds, steps, True)
    same_v = va == vb
    same_p = all(torch.equal(x, y) for x, y in zip(pa, pb))
    fin = all(bool(torch.isfinite(x).all()) for x in pa)
    print(f'{ds}: {steps} steps (each loop includes its own graph captures) free-running {ta * 1e3:.3f} ms/step, synchronised after every call {tb * 1e3:.3f} ms/step; loss values identical: {same_v}; '
          f'final parameters bitwise identical: {same_p}; finite: {fin}; first / last total {sum(va[0]):.5f} / {sum(va[-1]):.5f}')
    assert same_v and same_p and fin
