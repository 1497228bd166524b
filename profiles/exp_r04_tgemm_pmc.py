"""The NBA step's largest products alone (forward 7392 x 256 -> 512, the layer's backward launch), 40 launches each, for rocprofv3 counters."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sttode_amd import capi
dev = torch.device('cuda')
cols, J, I = 7392, 256, 512
X = torch.randn(cols, J, device=dev); W = torch.randn(I, J, device=dev); b = torch.randn(I, device=dev); Y = torch.empty(cols, I, device=dev)
dY = torch.randn(cols, I, device=dev); dX = torch.empty(cols, J, device=dev); gW = torch.zeros(I, J, device=dev); gb = torch.zeros(I, device=dev)
scratch = torch.empty(8 << 20, device=dev)
st = capi.stream_ptr()
for _ in range(40):
    capi.call('sttode_tlinear', X, J, 1, W, J, 0, b, None, 0, Y, I, cols, J, I, 1, 0, st)
for _ in range(40):
    capi.call('sttode_tlinear_bwd', dY, I, W, J, Y, I, dX, J, J, 0, X, J, 1, gW, J, gb, cols, I, J, scratch, scratch.numel(), st)
torch.cuda.synchronize()
