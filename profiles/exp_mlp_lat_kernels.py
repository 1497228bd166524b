"""Stand-alone timings of sttode_mlp_block0 / sttode_mlp_block1 (ETH shapes) in their latency and throughput forms over column counts."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sttode_amd import capi, packing
from sttode_amd.weights import make_weights
dev = torch.device('cuda:0')
sd = make_weights(1234)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
P0 = packing.pack_block(sd, 0, 8, 12, first=True)
P1 = packing.pack_block(sd, 1, 8, 12, first=False)
s0, s1 = t(P0['stream']), t(P1['stream'])
K = 20
def timeit(go, reps=50):
    for _ in range(5):
        go()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        go()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for n in [int(x) for x in os.environ.get("NAGENTS", "1,8,32,64").split(",")]:
    ncols = n * K
    A0x, A0y, A1y = (torch.randn(n, 512, device=dev) for _ in range(3))
    z, xpad = torch.randn(ncols, 32, device=dev), torch.randn(n, 16, device=dev)
    dbuf, ybuf = torch.zeros(ncols, 16, device=dev), torch.zeros(ncols, 32, device=dev)
    st1, cur, orig = torch.randn(ncols, 96, device=dev), torch.randn(n, 2, device=dev), torch.randn(n, 2, device=dev)
    pred = torch.zeros(ncols, 24, device=dev)
    for tiles in (1 << 30, 0):
        capi.call('sttode_set_latency_tiles', -1, tiles, -1)
        u0 = timeit(lambda: capi.call('sttode_mlp_block0', A0x, A0y, s0, int(P0['n_chunks']), z, xpad, dbuf, ybuf, ncols, K, 1, 2, capi.stream_ptr()))
        u1 = timeit(lambda: capi.call('sttode_mlp_block1', A1y, s1, int(P1['n_chunks']), z, st1, ybuf, cur, orig, pred, ncols, K, 12, 2, capi.stream_ptr()))
        print(f'ncols {ncols:5d} {"latency   " if tiles else "throughput"} form: mlp_block0 {u0:6.1f} us  mlp_block1 {u1:6.1f} us   checksum {float(pred.sum()) + float(dbuf.sum()):.4f}', flush=True)
