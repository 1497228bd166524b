"""Phase stamps of the one-launch scene form (csrc/scene_lat.hip built with -DSL_DIAG_TRACE -> sttode_amd/lib/variants/lib_sltrace.so):
who waits for whom inside the launch.  Build:  cd sttode_amd/csrc && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DSL_DIAG_TRACE -c scene_lat.hip
-o /tmp/t.o && hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/variants/lib_sltrace.so /tmp/t.o $(ls build/*.o | grep -v scene_lat.o)
Run:  STTODE_HIP_LIB=sttode_amd/lib/variants/lib_sltrace.so python profiles/exp_r03_scene_trace.py"""
import os, sys, ctypes, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(sys.path[0], 'tests'))
from helpers import make_args
from sttode_amd import STTODENet, scenes, capi
from sttode_amd.weights import make_weights, to_torch_state_dict
dev = torch.device('cuda')
m = STTODENet(make_args('eth', 8, 12), dev).eval()
m.load_state_dict(to_torch_state_dict(make_weights(1234)))
L = capi.lib()
L.sttode_scene_debug_buffer.argtypes = [ctypes.c_void_p]
for sid in (300001, 300007, 300013):
    o, p = scenes.eth_scene(sid)
    n = o.shape[0]
    A, C = (n + 15) // 16, (n * 20 + 15) // 16
    grid = 2 * A + 2 * C
    dbg = torch.zeros(grid * 8, dtype=torch.int64, device=dev)
    L.sttode_scene_debug_buffer(ctypes.c_void_p(dbg.data_ptr()))
    reps = []
    for it in range(6):
        dbg.zero_()
        m.set_data(None, torch.from_numpy(o), torch.from_numpy(p), None, None)
        m.inference(None)
        torch.cuda.synchronize()
        reps.append(dbg.cpu().numpy().reshape(grid, 8).copy())
    d = reps[-1].astype(np.float64)
    t0 = d[:, 0].min()
    us = lambda x: (x - t0) / 100.0
    print(f'scene {sid}: n = {n} agents, {C} trajectory tiles, grid {grid}; times in us after the first workgroup started')
    names = {0: 'E', 1: 'G'}
    for b in range(grid):
        r = d[b]
        if b < 2 * A:
            if b % 2 == 0:
                print(f'  E{b // 2}: start {us(r[0]):6.1f} | embed done {us(r[1]):6.1f} | pf published {us(r[2]):6.1f} | A1y published {us(r[3]):6.1f}')
            else:
                print(f'  G{b // 2}: start {us(r[0]):6.1f} | front-end done {us(r[1]):6.1f} | state0 published {us(r[2]):6.1f}')
        else:
            t = (b - 2 * A) // 2
            if (b - 2 * A) % 2 == 0:
                if t < 2 or t == C - 1:
                    print(f'  Y{t}: start {us(r[0]):6.1f} | flags seen {us(r[1]):6.1f} | layer-1 table {us(r[2]):6.1f} | ybuf published {us(r[3]):6.1f}')
            elif t < 2 or t == C - 1:
                print(f'  X{t}: start {us(r[0]):6.1f} | flags seen {us(r[1]):6.1f} | layer-1 table {us(r[2]):6.1f} | d ready {us(r[3]):6.1f} | GRU done {us(r[4]):6.1f} | '
                      f'y/A1y seen {us(r[5]):6.1f} | pred written {us(r[6]):6.1f} | core clock {r[7] / ((r[6] - r[0]) / 100.0) / 1e3:.2f} GHz')
    ends = [max(x[:, 1:].max() for x in [rr]) - rr[:, 0].min() for rr in reps]
    print('  launch span (first start -> last stamp), us, 6 calls:', ' '.join(f'{e / 100.0:.1f}' for e in ends))
