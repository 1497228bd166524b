"""Per-phase in-kernel stamps of the fused chain kernel (diagnostic build lib_stamps.so, C32_DIAG_STAMPS): where a group's time goes.
    STTODE_HIP_LIB=sttode_amd/lib/variants/lib_stamps.so python profiles/exp_chain_stamps.py 128 512"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import make_args
from sttode_amd import STTODENet, capi, scenes
from sttode_amd.weights import make_weights, to_torch_state_dict
dev = torch.device('cuda:0')
m = STTODENet(make_args('eth', 8, 12), dev).eval()
m.load_state_dict(to_torch_state_dict(make_weights(1234)), strict=True)
L = capi.lib()
for S in [int(a) for a in sys.argv[1:]] or [128]:
    sb = scenes.make_scene_batch(range(S), 'eth')
    m.set_scene_batch(sb.past, sb.future, sb.scene_ptr)
    m.native().set_chain(1)
    for _ in range(3):
        m.inference(None)
    ngroups = (sb.n_agents * 20 + 127) // 128
    nwg = max(512, ngroups + (sb.n_agents + 15) // 16)   # one workgroup per group (+ the roles of a fused launch: group blocks come after them)
    dbg = torch.zeros(nwg * 4 * 16, dtype=torch.int64, device=dev)
    L.sttode_chain_debug_buffer.argtypes = [ctypes.c_void_p]
    L.sttode_chain_debug_buffer(dbg.data_ptr())
    m.inference(None)
    torch.cuda.synchronize()
    L.sttode_chain_debug_buffer(None)
    d = dbg.cpu().numpy().reshape(nwg, 4, 8, 2)
    valid = d[:, :, 0, 0] > 0
    cyc = d[..., 0].astype(np.float64); ns = d[..., 1].astype(np.float64) * 10.0
    names = ['mlp0x', 'mlp0y', 'gru', 'mlp1']
    print(f'scenes {S}: {ngroups} groups, {int(valid.sum())} stamped group passes')
    for k in range(4):
        dc = (cyc[:, :, k + 1] - cyc[:, :, k])[valid]; dn = (ns[:, :, k + 1] - ns[:, :, k])[valid]
        print(f'  {names[k]:6s} cycles median {np.median(dc):10.0f} min {dc.min():10.0f} max {dc.max():10.0f} | us median {np.median(dn) / 1e3:8.1f}  -> clock {np.median(dc) / np.median(dn):.2f} GHz')
    tot = (ns[:, :, 4] - ns[:, :, 0])[valid]
    t0 = ns[:, 0, 0][valid[:, 0]]
    print(f'  group total us median {np.median(tot) / 1e3:.1f} min {tot.min() / 1e3:.1f} max {tot.max() / 1e3:.1f}; first-group start spread {(t0.max() - t0.min()) / 1e3:.1f} us; '
          f'kernel span {(ns[:, :, 4][valid].max() - t0.min()) / 1e3:.1f} us')
    b3 = os.environ.get('STTODE_BF16X3', '0') not in ('', '0')
    per_tile = 12 * 32 if b3 else 16 * 64      # matrix-pipe cycles per 32 x 32 x 32 tile: six bf16 MFMAs per k block vs sixteen fp32 MFMAs
    ideal = {'mlp0x': 152 * per_tile, 'mlp0y': 152 * per_tile, 'gru': 8 * 37 * per_tile, 'mlp1': 200 * per_tile}
    print('  MFMA-only cycles per phase (' + ('bf16x3' if b3 else 'f32') + '):', ideal)
