"""Where the HOST time of the one-scene-per-call loop goes (test.py:171-188 pattern): cProfile over set_data + inference + .cpu()."""
import os, sys, time, cProfile, pstats, io, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(sys.path[0], 'tests'))
from helpers import make_args
from sttode_amd import STTODENet, scenes
from sttode_amd.weights import make_weights, to_torch_state_dict
dev = torch.device('cuda')
m = STTODENet(make_args('eth', 8, 12), dev).eval()
m.load_state_dict(to_torch_state_dict(make_weights(1234)))
data = [scenes.eth_scene(300000 + i) for i in range(256)]
data = [(torch.from_numpy(o), torch.from_numpy(p)) for o, p in data]
def loop(reps):
    for _ in range(reps):
        for o, p in data:
            m.set_data(None, o, p, None, None)
            out = m.inference(None)
            out = out.cpu()
loop(1)
t = time.perf_counter(); loop(4); dt = time.perf_counter() - t
print(f'{dt / (4 * len(data)) * 1e3:.3f} ms/scene')
pr = cProfile.Profile(); pr.enable(); loop(4); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28); print(s.getvalue()[:6000])
