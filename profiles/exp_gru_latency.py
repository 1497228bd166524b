"""gru_cols: latency form (one 16-column tile per workgroup, hidden units split over six waves) vs throughput form (a tile per wave,
weights resident in LDS) over column counts.  STTODE_GRU_LAT_TILES selects: run twice,
    STTODE_GRU_LAT_TILES=0 python profiles/exp_gru_latency.py ; STTODE_GRU_LAT_TILES=100000000 python profiles/exp_gru_latency.py"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sttode_amd import capi, packing
from sttode_amd.weights import make_weights
dev = torch.device('cuda:0')
sd = make_weights(1234)
P = packing.pack_block(sd, 1, 8, 12, first=False)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
W = [t(P[k]) for k in ('convP', 'convB', 'wihP', 'whhP', 'gbias')]
for ncols in (160, 640, 2048, 4096, 8192, 16384, 32768, 65536, 172900):
    x = torch.randn(ncols, 16, device=dev)
    st = torch.zeros(ncols, 96, device=dev)
    go = lambda: capi.call('sttode_gru_cols', x, *W, st, ncols, 8, 1, capi.stream_ptr())
    for _ in range(3):
        go()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        go()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print(f'LAT_TILES={os.environ.get("STTODE_GRU_LAT_TILES", "default")}: ncols {ncols:7d} {us:8.1f} us  {ncols * 592896 / us / 1e6:6.1f} TFLOP/s  checksum {float(st.sum()):.4f}')
