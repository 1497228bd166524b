import ctypes, os, sys, numpy as np, torch
ROOT='/root/repo'
L = ctypes.CDLL(os.path.join(os.environ.get('GRAFT_REPO_ROOT', ROOT), 'sttode_amd', 'lib', 'libsttode_diag.so'))
L.sttode_diag_last_error.restype = ctypes.c_char_p
P, I, D = ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_double)
L.sttode_diag_stream.argtypes = [I, I, I, I, P, ctypes.c_long, P, D, P]
scr = torch.zeros(1024 * 1024, device='cuda'); blob = torch.randn(1024 * 1024, device='cuda')
for wg in (1, 2):
    vals = {3: [], 4: [], 5: [], 1: []}
    for _ in range(5):
        for k in vals:
            tf = ctypes.c_double()
            n = 512 // (3 if k <= 2 else 9)
            rc = L.sttode_diag_stream(k, wg, n, 5, blob.data_ptr(), blob.numel(), scr.data_ptr(), ctypes.byref(tf), None)
            assert rc == 0, L.sttode_diag_last_error()
            vals[k].append(tf.value)
    for k in vals: print('WG/CU', wg, 'shape', k, 'median %.1f best %.1f' % (np.median(vals[k]), max(vals[k])))
