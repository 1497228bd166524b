"""Drop-in call pattern of test.py:171-199: one scene per call (set_data + inference + D2H of the futures), unbatched."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(sys.path[0], 'tests'))
from helpers import make_args
from sttode_amd import STTODENet, scenes
from sttode_amd.weights import make_weights, to_torch_state_dict
dev = torch.device('cuda')
m = STTODENet(make_args('eth', 8, 12), dev).eval()
m.load_state_dict(to_torch_state_dict(make_weights(1234)))
data = [scenes.eth_scene(300000 + i) for i in range(256)]
data = [(torch.from_numpy(o), torch.from_numpy(p)) for o, p in data]
def run(sync_each):
    tot = 0
    for o, p in data:
        m.set_data(None, o, p, None, None)          # host tensors in, as the reference's loader hands them over
        out = m.inference(None)
        if sync_each:
            out = out.cpu()                          # test.py:186-188 moves every prediction to NumPy
        tot += o.shape[0] * 20
    torch.cuda.synchronize()
    return tot
run(True)
for sync_each in (True, False):
    t = time.perf_counter(); tot = run(sync_each); dt = time.perf_counter() - t
    print(f'per-scene loop, D2H each call={sync_each}: {dt / len(data) * 1e3:.3f} ms/scene, {tot / dt / 1e6:.2f} M traj/s')
# host split of the loop with the D2H of every prediction
T = [0.0, 0.0, 0.0]
for o, p in data:
    t0 = time.perf_counter(); m.set_data(None, o, p, None, None)
    t1 = time.perf_counter(); out = m.inference(None)
    t2 = time.perf_counter(); out = out.cpu()
    t3 = time.perf_counter()
    T[0] += t1 - t0; T[1] += t2 - t1; T[2] += t3 - t2
print('host split per scene (ms): set_data %.3f, inference (enqueue) %.3f, .cpu() (waits for the GPU + D2H) %.3f' % tuple(1e3 * x / len(data) for x in T))
