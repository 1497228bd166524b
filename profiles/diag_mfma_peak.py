import ctypes, torch, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sttode_amd import capi
L = capi.lib()
scr = torch.zeros(256 * 1024, device='cuda')
for w in (4, 8, 16, -4, -8, -16):
    for it in (2000, 20000):
        tf = ctypes.c_double()
        rc = L.sttode_diag_mfma_peak(w, it, 20, ctypes.c_void_p(scr.data_ptr()), ctypes.byref(tf), None)
        print('waves/CU', w, 'iters', it, 'rc', rc, 'TFLOP/s %.1f' % tf.value)
print('--- instruction / operand shapes (sttode_diag_mfma_kinds): 0 16x16x4 reg | 1 16x16x4+LDS | 2 32x32x2 reg | 3 32x32x2+LDS')
for w in (8, 12, 16):
    for kind in (0, 1, 2, 3):
        tf = ctypes.c_double()
        rc = L.sttode_diag_mfma_kinds(kind, w, 20000, 40, ctypes.c_void_p(scr.data_ptr()), ctypes.byref(tf), None)
        print('waves/CU', w, 'kind', kind, 'rc', rc, 'TFLOP/s %.1f' % tf.value)
