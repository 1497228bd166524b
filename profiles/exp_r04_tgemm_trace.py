"""Where the time of ONE training GEMM launch goes: per-workgroup stamps (100 MHz clock) at start, first tile in LDS, reduction done, end.
Forward 7392 x 256 -> 512 (928 workgroups, one round on 1024 slots) and 7392 x 512 -> 256."""
import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sttode_amd import capi
dev = torch.device('cuda')
L = capi.lib()
L.sttode_tgemm_debug_buffer.argtypes = [ctypes.c_void_p]
st = capi.stream_ptr()
for cols, J, I in ((7392, 256, 512), (7392, 512, 256)):
    X = torch.randn(cols, J, device=dev); W = torch.randn(I, J, device=dev); b = torch.randn(I, device=dev); Y = torch.empty(cols, I, device=dev)
    nwg = ((cols + 63) // 64) * ((I + 63) // 64)
    dbg = torch.zeros(nwg * 4, dtype=torch.int64, device=dev)
    for _ in range(5):
        capi.call('sttode_tlinear', X, J, 1, W, J, 0, b, None, 0, Y, I, cols, J, I, 1, 0, st)
    torch.cuda.synchronize()
    for rep in range(3):
        L.sttode_tgemm_debug_buffer(ctypes.c_void_p(dbg.data_ptr()))
        capi.call('sttode_tlinear', X, J, 1, W, J, 0, b, None, 0, Y, I, cols, J, I, 1, 0, st)
        L.sttode_tgemm_debug_buffer(None)
        torch.cuda.synchronize()
        d = dbg.cpu().numpy().reshape(nwg, 4).astype(np.float64) * 0.01      # us
        t0 = d[:, 0].min()
        d -= t0
        q = lambda x: '%.1f / %.1f / %.1f' % (np.percentile(x, 5), np.median(x), np.percentile(x, 95))
        print(f'{cols}x{J}->{I} rep {rep}: {nwg} workgroups, launch {d[:, 3].max():.1f} us; start (5/50/95 %) {q(d[:, 0])}; '
              f'first tile in LDS after {q(d[:, 1] - d[:, 0])}; reduction {q(d[:, 2] - d[:, 1])}; epilogue {q(d[:, 3] - d[:, 2])}; end {q(d[:, 3])}')
