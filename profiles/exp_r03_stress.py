"""Stress of the in-launch hand-off under the product's own load: N pipelined fused calls (three in flight, workspace slots reused
every fourth call, three input variants of one shape in rotation, fixed latents per variant), every call's predictions compared BITWISE
with the serial reference of its variant and checked for NaN; the time-out word of every slot must stay 0.
    python profiles/exp_r03_stress.py [calls=3000] [scenes=512] [mode=f32|bf16x3]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import make_args
from sttode_amd import STTODENet, scenes
from sttode_amd.weights import make_weights, to_torch_state_dict
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 512
mode = sys.argv[3] if len(sys.argv) > 3 else 'f32'
dev = torch.device('cuda:0')
m = STTODENet(make_args('eth', 8, 12), dev).eval()
m.load_state_dict(to_torch_state_dict(make_weights(1234)), strict=True)
m.mfma_mode = mode
sb = scenes.make_scene_batch(range(S), 'eth')
n = sb.n_agents
var = []
for v in range(3):
    past = torch.from_numpy((sb.past * (1.0 + 0.02 * v) + 0.05 * v).astype(np.float32)).to(dev)
    var.append((past, torch.from_numpy(sb.future).to(dev), torch.from_numpy(sb.scene_ptr).to(dev), torch.from_numpy(scenes.latents(9 + v, n)).to(dev)))
ref = []
for past, fut, ptr, z in var:
    m.set_scene_batch(past, fut, ptr)
    ref.append(m.inference(None, z=z).clone())
torch.cuda.synchronize()
bad, pend, t0 = 0, [], time.perf_counter()
for i in range(calls):
    v = (i * 7 + i // 5) % 3
    past, fut, ptr, z = var[v]
    m.set_scene_batch(past, fut, ptr)
    pend.append((v, m.inference_async(z=z)))
    if len(pend) >= 3:
        vv, h = pend.pop(0)
        out = m.wait(h)
        if not torch.equal(out, ref[vv]):
            bad += 1
            print(f'call {i - 2}: MISMATCH (variant {vv}), nan={bool(torch.isnan(out).any())}', flush=True)
    if i % 500 == 499:
        print(f'{i + 1} calls, {bad} mismatches, {1e3 * (time.perf_counter() - t0) / (i + 1):.3f} ms/call', flush=True)
while pend:
    vv, h = pend.pop(0)
    if not torch.equal(m.wait(h), ref[vv]):
        bad += 1
torch.cuda.synchronize()
tmo = 0
for (nn, SS, slot), (buf, pred) in m._async_bufs.items():
    off, _ = m.native().layout(nn, SS)
    tmo += int(buf[off['flags'] + (nn + 15) // 16].view(torch.int32))
print(f'{calls} pipelined fused calls ({mode}, {S} scenes, {n * 20} trajectories each): {bad} mismatches, time-out words set: {tmo}, '
      f'{1e3 * (time.perf_counter() - t0) / calls:.3f} ms/call')
sys.exit(1 if bad or tmo else 0)
