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
    m.train()
    opt = Adam(m.parameters(), lr=1e-4)
    vals = []
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(steps):
        d = data[i % len(data)]
        if ds == 'eth':
            m.set_data(None, d[0], d[1], None, None)
        else:
            m.set_data_nba(d)
        out = m.forward()
        if sync: torch.cuda.synchronize()
        opt.zero_grad(); out[0].backward(); opt.step()
        if sync: torch.cuda.synchronize()
        vals.append(out[1:])
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / steps
    return vals, [p.detach().clone() for p in m.parameters()], dt


for ds, steps in (('eth', 1200), ('nba', 300)):
    va, pa, ta = loop(ds, steps, False)
    vb, pb, tb = loop(ds, steps, True)
    same_v = va == vb
    same_p = all(torch.equal(x, y) for x, y in zip(pa, pb))
    fin = all(bool(torch.isfinite(x).all()) for x in pa)
    print(f'{ds}: {steps} steps free-running {ta * 1e3:.3f} ms/step, synchronised after every call {tb * 1e3:.3f} ms/step; loss values identical: {same_v}; '
          f'final parameters bitwise identical: {same_p}; finite: {fin}; first / last total {sum(va[0]):.5f} / {sum(va[-1]):.5f}')
    assert same_v and same_p and fin
