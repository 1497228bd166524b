"""Stress of the one-launch scene form's in-launch hand-offs (csrc/scene_lat.hip: E / G / Y / X roles, sc1 payloads, flags): N one-scene calls
in the evaluation loop's pattern (set_data + inference, fixed latents per scene), each compared BITWISE with the six-launch form of the
same scene computed up front; NaN anywhere or a set time-out word fails.
    python profiles/exp_r03_scene_stress.py [calls=6000] [scenes=384]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import make_args
from sttode_amd import STTODENet, scenes
from sttode_amd.weights import make_weights, to_torch_state_dict
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 384
dev = torch.device('cuda:0')
m = STTODENet(make_args('eth', 8, 12), dev).eval()
m.load_state_dict(to_torch_state_dict(make_weights(1234)), strict=True)
data = []
for s in range(S):
    o, p = scenes.eth_scene(700000 + s)
    data.append((torch.from_numpy(o), torch.from_numpy(p), torch.from_numpy(scenes.latents(s, o.shape[0])).to(dev)))
m.native().set_scene_launch(0)
ref = []
for o, p, z in data:
    m.set_data(None, o, p, None, None)
    ref.append(m.inference(None, z=z).clone())
m.native().set_scene_launch(-1)
torch.cuda.synchronize()
bad, t0 = 0, time.perf_counter()
for i in range(calls):
    s = (i * 131 + i // 7) % S
    o, p, z = data[s]
    m.set_data(None, o, p, None, None)
    out = m.inference(None, z=z)
    if not torch.equal(out, ref[s]):
        bad += 1
        print(f'call {i}: MISMATCH (scene {s}, {o.shape[0]} agents), nan={bool(torch.isnan(out).any())}', flush=True)
    if i % 1000 == 999:
        print(f'{i + 1} calls, {bad} mismatches, {1e3 * (time.perf_counter() - t0) / (i + 1):.3f} ms/call', flush=True)
tmo = 0
for (nn, SS), (buf, off) in m._wscache.items():
    tmo += int(buf[off['flags'] + (nn + 15) // 16].view(torch.int32))
print(f'{calls} one-scene calls (one-launch form) over {S} scenes of 2-32 agents: {bad} mismatches against the six-launch form, time-out words set: {tmo}')
sys.exit(1 if bad or tmo else 0)
