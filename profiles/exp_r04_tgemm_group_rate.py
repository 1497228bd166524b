"""Grouped launches of the NBA step's largest products (decoder_x / decoder_y of a block: same input, separate weights): microseconds and
fraction of the fp32-MFMA peak (157.3 TFLOP/s) per LAUNCH -- forward pair 7392 x 256 -> 512, 7392 x 512 -> 256; backward pairs (dX + dW
of both layers: four products) -- against the same products one per launch."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sttode_amd import capi
dev = torch.device('cuda')
cols = 7392
st = capi.stream_ptr()
scratch = torch.empty(8 << 20, device=dev)
buf = torch.empty(32 << 20, device=dev)
def timed(fn, reps=40):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for J, I in ((256, 512), (512, 256)):
    X = torch.randn(cols, J, device=dev)
    W = [torch.randn(I, J, device=dev) / 16 for _ in range(2)]; b = [torch.randn(I, device=dev) for _ in range(2)]
    Y = [torch.empty(cols, I, device=dev) for _ in range(2)]
    dY = [torch.randn(cols, I, device=dev) for _ in range(2)]; dX = [torch.empty(cols, J, device=dev) for _ in range(2)]
    gW = [torch.zeros(I, J, device=dev) for _ in range(2)]; gb = [torch.zeros(I, device=dev) for _ in range(2)]
    def fwd(group):
        if group: capi.call('sttode_tgemm_group', 1)
        for i in range(2): capi.call('sttode_tlinear', X, J, 1, W[i], J, 0, b[i], None, 0, Y[i], I, cols, J, I, 1, 0, st)
        if group: capi.call('sttode_tgemm_group', 0)
    def bwd(group):
        capi.call('sttode_twgrad_defer', 1, buf, buf.numel())
        if group: capi.call('sttode_tgemm_group', 1)
        for i in range(2):
            capi.call('sttode_tlinear_bwd', dY[i], I, W[i], J, Y[i], I, dX[i], J, J, 0, X, J, 1, gW[i], J, gb[i], cols, I, J, scratch, scratch.numel(), st)
        if group: capi.call('sttode_tgemm_group', 0)
        capi.call('sttode_twgrad_defer', 0, None, 0)
    ff, fb = 2 * 2.0 * cols * J * I, 2 * 4.0 * cols * J * I
    for name, fn, fl in (('forward pair', fwd, ff), ('backward pair (4 products + their reduction)', bwd, fb)):
        t1, tg = timed(lambda: fn(False)), timed(lambda: fn(True))
        print(f'{cols} x {J} -> {I}  {name}: one product (pair) per launch {t1:.1f} us = {fl / t1 / 157.3e6:.2f} of peak; grouped {tg:.1f} us = {fl / tg / 157.3e6:.2f} of peak')
