"""Exploratory (round 3): does a THREE-WAY bf16 split on the bf16 matrix cores beat the fp32 MFMA in the chain's weight-stream structure?
libsttode_diag.so shapes 6 / 7 (csrc/diag/diag.hip: diag_stream_b3_kernel) against shape 2 (the fused chain's structure: 32x32x2 fp32,
12 KiB chunks, fragment prefetch).  TFLOP/s are fp32-EQUIVALENT (65 536 FLOP per 32 x 32 x 32 tile and wave in every shape).
    python profiles/exp_r03_bf16x3_probe.py [out.json]"""
import ctypes, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = ctypes.CDLL(os.path.join(ROOT, 'sttode_amd', 'lib', 'libsttode_diag.so'))
L.sttode_diag_last_error.restype = ctypes.c_char_p
P, I, D = ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_double)
L.sttode_diag_stream.argtypes = [I, I, I, I, P, ctypes.c_long, P, D, P]
scr = torch.zeros(1024 * 1024, device='cuda')
blob = (torch.randn(1536 * 1024, device='cuda') * 0.01).contiguous()
SN = {2: 'fp32 32x32x2 stream, 12 KiB chunks, fragment prefetch (the chain)', 6: 'bf16 x3 split, 18 KiB chunks, 256 VGPRs (2 waves / SIMD)',
      7: 'bf16 x3 split, 18 KiB chunks, 512 VGPRs (1 wave / SIMD)'}
res = {}
for wg in (1, 2):
    vals = {k: [] for k in SN}
    for _ in range(5):
        for k in SN:
            if k == 7 and wg == 2:
                continue
            tf = ctypes.c_double()
            rc = L.sttode_diag_stream(k, wg, 512 // 3, 5, blob.data_ptr(), blob.numel(), scr.data_ptr(), ctypes.byref(tf), None)
            assert rc == 0, L.sttode_diag_last_error()
            vals[k].append(tf.value)
    for k in SN:
        if vals[k]:
            res[f'{SN[k]} @ {wg} WG/CU'] = {'median_tflops_fp32_equivalent': float(np.median(vals[k])), 'best': max(vals[k])}
            print(f'WG/CU {wg}  {SN[k]:70s} median {np.median(vals[k]):6.1f}  best {max(vals[k]):6.1f} TFLOP/s (fp32-equivalent)', flush=True)
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], 'w'), indent=1)
