"""Whole single-scene call (model.inference of one scene, no D2H) with the latency forms of gru_cols / mlp_block0 / mlp_block1 on and
off, per scene size; then the per-stage timings of the native pipeline for one 32-pedestrian scene."""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import make_args
from sttode_amd import STTODENet, capi, scenes
from sttode_amd.weights import make_weights, to_torch_state_dict
dev = torch.device('cuda')
m = STTODENet(make_args('eth', 8, 12), dev).eval()
m.load_state_dict(to_torch_state_dict(make_weights(1234)))
for n in [int(x) for x in os.environ.get("NAGENTS", "2,8,32,64,128").split(",")]:
    o, p = scenes.eth_scene(777 + n, n_min=n, n_max=n)
    o, p = torch.from_numpy(o), torch.from_numpy(p)
    for gt, mt in (((0, 0), (1 << 30, 0), (1 << 30, 1 << 30)) if not os.environ.get("LAT_ONLY") else ((1 << 30, 1 << 30),)):
        capi.call('sttode_set_latency_tiles', gt, mt, mt)
        m.set_data(None, o, p, None, None)
        z = torch.randn(n * 20, 32, device=dev)
        for _ in range(5):
            m.inference(None, z=z)
        torch.cuda.synchronize()
        reps = 200
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(reps):
            m.inference(None, z=z)
        e1.record()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        print(f'n={n:4d} ({n * 20:5d} traj) lat(gru,mlp)=({int(gt > 0)},{int(mt > 0)}): GPU {e0.elapsed_time(e1) / reps * 1e3:7.1f} us/call, '
              f'host enqueue {(t1 - t0) / reps * 1e6:6.1f} us/call', flush=True)
