"""MFMA-shape and weight-stream probes (sttode_amd/lib/libsttode_diag.so, NOT the product library).

    python profiles/diag_mfma_peak.py [out.json]

Part 1: bare issue loops of v_mfma_f32_16x16x4_f32 / v_mfma_f32_32x32x2_f32 (registers only, or one ds_read_b128 A fragment per
4 MFMAs).  Part 2: the decoder-MLP weight-stream structure (LDS-DMA double buffer, one barrier per chunk) in both shapes.
Variants are interleaved over several rounds in ONE process (cdna_hip_programming.md rule 24); median and best are reported.
"""
import ctypes
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = ctypes.CDLL(os.path.join(ROOT, 'sttode_amd', 'lib', 'libsttode_diag.so'))
L.sttode_diag_last_error.restype = ctypes.c_char_p
P, I, D = ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_double)
L.sttode_diag_mfma_kinds.argtypes = [I, I, I, I, P, D, P]
L.sttode_diag_stream.argtypes = [I, I, I, I, P, ctypes.c_long, P, D, P]

scr = torch.zeros(1024 * 1024, device='cuda')
blob = torch.randn(1024 * 1024, device='cuda')  # 4 MiB of random weights: L2/MALL resident like the packed model
res = {'kinds': {}, 'stream': {}}
KN = {0: '16x16x4 reg', 1: '16x16x4 + ds_read_b128/4 MFMA', 2: '32x32x2 reg', 3: '32x32x2 + ds_read_b128/4 MFMA'}
ROUNDS = 5
for w in (4, 8, 12, 16):
    vals = {k: [] for k in KN}
    for _ in range(ROUNDS):
        for k in KN:
            tf = ctypes.c_double()
            rc = L.sttode_diag_mfma_kinds(k, w, 20000, 20, scr.data_ptr(), ctypes.byref(tf), None)
            assert rc == 0, L.sttode_diag_last_error()
            vals[k].append(tf.value)
    for k in KN:
        res['kinds'][f'{KN[k]} @ {w} waves/CU'] = {'median_tflops': float(np.median(vals[k])), 'best_tflops': max(vals[k])}
        print(f'waves/CU {w:2d}  {KN[k]:32s} median {np.median(vals[k]):6.1f}  best {max(vals[k]):6.1f} TFLOP/s', flush=True)

SN = {0: '16x16x4 stream, 18 KiB chunks (round-1 mlp_block0 shape)', 1: '32x32x2 stream, 12 KiB chunks', 2: '32x32x2 stream, 12 KiB chunks, fragment prefetch',
      3: '32x32x2 stream, 36 KiB chunks', 4: '32x32x2 stream, 36 KiB chunks, fragment prefetch',
      5: '32x32x2 stream, 36 KiB chunks, two interleaved accumulator chains'}
for wg in (1, 2, 3):
    vals = {k: [] for k in SN}
    for _ in range(ROUNDS):
        for k in SN:
            if wg > (3 if k == 0 else 2):
                continue
            n = 2048 if k == 0 else 512 // (3 if k <= 2 else 9)   # equal FLOP per workgroup (1.2 GFLOP)
            tf = ctypes.c_double()
            rc = L.sttode_diag_stream(k, wg, n, 5, blob.data_ptr(), blob.numel(), scr.data_ptr(), ctypes.byref(tf), None)
            assert rc == 0, L.sttode_diag_last_error()
            vals[k].append(tf.value)
    for k in SN:
        if vals[k]:
            res['stream'][f'{SN[k]} @ {wg} WG/CU'] = {'median_tflops': float(np.median(vals[k])), 'best_tflops': max(vals[k])}
            print(f'WG/CU {wg}  {SN[k]:60s} median {np.median(vals[k]):6.1f}  best {max(vals[k]):6.1f} TFLOP/s', flush=True)
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], 'w'), indent=1)
