"""Stand-alone operators over the C ABI (op-level drop-ins for the reference's functions).

mhgsa(...)  ==  Hyp_mhsa.forward / hyp_mhsa() (hyptransformerlib.py:29-311,403-454) for embed_dim = 64, 8 heads,
no masks / bias_kv / zero_attn / dropout (the only configuration the reference instantiates,
hypertransformer.py:29, model/STTODE.py:190-194).
"""
import numpy as np
import torch

from . import capi, packing

_PK_CACHE = {}


def _pk16_cached(w):
    """Fragment-ordered copy of a weight, cached per (storage address, version).  The entry keeps ``w`` alive, so the address
    cannot be recycled for a different tensor while the entry exists."""
    key = (w.data_ptr(), w._version, tuple(w.shape), str(w.device))
    if key not in _PK_CACHE:
        if len(_PK_CACHE) > 64:
            _PK_CACHE.clear()
        _PK_CACHE[key] = (w, torch.from_numpy(packing.pk16(w.detach().cpu().numpy())).to(w.device))
    return _PK_CACHE[key][1]


_ACT = {None: 0, 'none': 0, 'relu': 1, 'tanh': 2}


def linear_cols(x, weight, bias=None, relu=False, act=None):
    """y = act(x @ weight.T + bias) on the MFMA column-chain kernel. x [cols, K] (K, N multiples of 16); act None|'relu'|'tanh'."""
    code = 1 if relu else _ACT[act]
    x = x.contiguous()
    cols, K = x.shape
    N = weight.shape[0]
    out = torch.empty(cols, N, dtype=torch.float32, device=x.device)
    capi.call('sttode_linear_cols', x, K, K, None, 0, 0, _pk16_cached(weight), bias.contiguous() if bias is not None else None,
              out, N, cols, N, code, capi.stream_ptr())
    return out


@torch.no_grad()
def mhgsa(query, key, value, in_proj_weight, in_proj_bias, out_proj_weight, out_proj_bias, num_heads=8, need_weights=False):
    """Multi-head geodesic self/cross attention. query [L,Nb,64], key/value [S,Nb,64] -> (out [L,Nb,64], weights [Nb,L,S] | None)."""
    if query.device.type != 'cuda':
        raise capi.SttodeError('mhgsa runs only on a HIP device (no CPU fallback)')
    L, Nb, E = query.shape
    S = key.shape[0]
    if E != 64 or num_heads != 8:
        raise NotImplementedError('mhgsa kernel is built for embed_dim=64, num_heads=8')
    if key.shape != value.shape or key.shape[1] != Nb:
        raise ValueError('key/value shape mismatch')
    W, b = in_proj_weight, in_proj_bias
    q = linear_cols(query.reshape(L * Nb, E), W[:E].contiguous(), b[:E])
    k = linear_cols(key.reshape(S * Nb, E), W[E:2 * E].contiguous(), b[E:2 * E])
    v = linear_cols(value.reshape(S * Nb, E), W[2 * E:].contiguous(), b[2 * E:])
    scale = float(E // num_heads) ** -0.5
    if L == S:
        # scores [S, L] used untransposed (hyptransformerlib.py:261-265): rows = keys, columns = queries
        R, C, rows, cols, rs, cs = k, q, S, L, 1.0, scale
    else:
        R, C, rows, cols, rs, cs = q, k, L, S, scale, 1.0
    attn = torch.empty(rows * Nb, E, dtype=torch.float32, device=query.device)
    rowsum = torch.empty(Nb * 8 * rows, dtype=torch.float32, device=query.device) if need_weights else None
    wout = torch.empty(Nb, rows, cols, dtype=torch.float32, device=query.device) if need_weights else None
    st = Nb * E
    capi.call('sttode_mhgsa_attn', R, C, v, attn, rowsum, wout, rows, cols, Nb, st, E, st, E, st, E, st, E, rs, cs, capi.stream_ptr())
    out = linear_cols(attn, out_proj_weight, out_proj_bias).view(rows, Nb, E)
    return out, wout
