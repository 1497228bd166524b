"""ctypes binding of the C-ABI shared library (include/sttode_hip.h).

The product path has NO fallback: if ``libsttode_hip.so`` is missing or a call fails, this raises.
Pointers are raw device addresses (``tensor.data_ptr()``); kernels are enqueued on the caller's
current HIP stream (``torch.cuda.current_stream().cuda_stream``).
"""
import ctypes
import os

_LIB = None
LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'lib', 'libsttode_hip.so')

_P, _I, _L, _F = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float

# name -> argtypes (mirrors include/sttode_hip.h; tests/test_capi_symbols.py checks header == table == .so)
SIGNATURES = {
    'sttode_abi_version': [],
    'sttode_frontend_scenes': [_P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P],
    'sttode_frontend_nba': [_P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P],
    'sttode_frontend_future': [_P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P],
    'sttode_embed_qkv': [_P] * 11 + [_P, _P, _P, _P, _I, _I, _P],
    'sttode_mhgsa_attn': [_P, _P, _P, _P, _P, _P, _I, _I, _I] + [_L] * 8 + [_F, _F, _P],
    'sttode_post_attn': [_P] * 14 + [_P, _P, _I, _P, _I, _F, _P],
    'sttode_gru_cols': [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    'sttode_linear_cols': [_P, _I, _I, _P, _I, _I, _P, _P, _P, _I, _I, _I, _I, _P],
    'sttode_mlp_block0': [_P] * 10 + [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    'sttode_mlp_block1': [_P] * 5 + [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    'sttode_best_of_k': [_P, _P, _I, _I, _I, _F, _P, _P, _P],
}


class SttodeError(RuntimeError):
    pass


def lib():
    """Load (once) and return the shared library; raise loudly if it is not built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise SttodeError(f'{LIB_PATH} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                              f'or `make -C sttode_amd/csrc`. There is no CPU fallback.')
        L = ctypes.CDLL(LIB_PATH)
        L.sttode_last_error.restype = ctypes.c_char_p
        L.sttode_last_error.argtypes = []
        for name, args in SIGNATURES.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = ctypes.c_int
        _LIB = L
    return _LIB


def _ptr(t):
    if t is None:
        return None
    return t.data_ptr() if hasattr(t, 'data_ptr') else int(t)


def call(name, *args):
    """Invoke an entry point; tensors are converted to device pointers; non-zero status raises."""
    L = lib()
    conv = [(_ptr(a) if (a is None or hasattr(a, 'data_ptr')) else a) for a in args]
    rc = getattr(L, name)(*conv)
    if rc != 0:
        raise SttodeError(f'{name} failed (status {rc}): {L.sttode_last_error().decode()}')


def stream_ptr():
    import torch
    return torch.cuda.current_stream().cuda_stream
