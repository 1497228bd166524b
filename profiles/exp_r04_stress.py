"""Stress of the LAGGED pipelined form under the product's own load (round 4): N pipelined calls, several in flight, workspace / latent /
metric slots reused, three input variants of one shape in rotation.  Per call: predictions BITWISE those of the first time its variant ran
in this form (the form is deterministic: nothing in it depends on timing) and within 2e-5 of the serial reference, NaN-free; fused
best-of-K ADE / FDE bitwise the first run's; every 7th call draws its own latents (device latents) and is checked against a serial call
fed the latents it reports; every 11th call is waited for right behind its submission (its groups become a launch of their own).
    python profiles/exp_r04_stress.py [calls=3000] [scenes=512] [mode=f32|bf16x3]"""
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
if S < 48:
    m.native().set_chain(1)
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


def close(a, b, tol=2e-5):
    return bool(((a - b).abs() <= tol + tol * b.abs()).all())


first, first_met = {}, {}
bad, dev_lat_checked, pend, t0 = 0, 0, [], time.perf_counter()


def finish(i, v, h, own_z):
    global bad, dev_lat_checked
    a, f = m.best_of_k_async(h)
    out = m.wait(h)
    if torch.isnan(out).any():
        bad += 1; print(f'call {i}: NaN', flush=True); return
    if own_z:                                       # the call drew its own latents: a serial call fed those latents must agree
        past, fut, ptr, _ = var[v]
        keep = out.clone()
        m.set_scene_batch(past, fut, ptr)
        ser = m.inference(None, z=h['z'])
        dev_lat_checked += 1
        if not close(keep, ser):
            bad += 1; print(f'call {i}: device-latent call differs from the serial call on its own latents', flush=True)
        return
    if v not in first:
        first[v], first_met[v] = out.clone(), (a.clone(), f.clone())
        if not close(out, ref[v]):
            bad += 1; print(f'call {i}: variant {v} beyond 2e-5 of the serial reference', flush=True)
        return
    if not torch.equal(out, first[v]) or not torch.equal(a, first_met[v][0]) or not torch.equal(f, first_met[v][1]):
        bad += 1; print(f'call {i}: MISMATCH with the first run of variant {v}', flush=True)


for i in range(calls):
    v = (i * 7 + i // 5) % 3
    past, fut, ptr, z = var[v]
    own_z = i % 7 == 3
    m.set_scene_batch(past, fut, ptr)
    h = m.inference_async(z=None if own_z else z, metrics_gt=fut)
    pend.append((i, v, h, own_z))
    if i % 11 == 5:                                 # wait right behind the call: nobody else has enqueued its groups
        finish(*pend.pop())
    while len(pend) > 3:
        finish(*pend.pop(0))
    if i % 500 == 499:
        print(f'{i + 1} calls, {bad} mismatches, {1e3 * (time.perf_counter() - t0) / (i + 1):.3f} ms/call', flush=True)
while pend:
    finish(*pend.pop(0))
torch.cuda.synchronize()
print(f'{calls} pipelined lagged calls ({mode}, {S} scenes, {n * 20} trajectories each): {bad} mismatches, {dev_lat_checked} device-latent calls '
      f'checked against serial calls on their latents, {1e3 * (time.perf_counter() - t0) / calls:.3f} ms/call (incl. the checks)')
sys.exit(1 if bad else 0)
