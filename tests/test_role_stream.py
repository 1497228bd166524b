"""CPU checks of the throughput-form role stream (sttode_amd.packing.role_stream, consumed by csrc/role32.hpp): the host-side foldings and
the tile / chunk ORDER, by a NumPy walk over the chunk program that mirrors the kernel's data flow tile by tile (32 x 32 blocks unpacked
from their MFMA fragment order) -- against the CPU oracle's PastEncoder, block-0 conv + GRU and layer-1 pre-activations.  No GPU."""
import numpy as np
import pytest
import torch

from helpers import make_args, oracle_model
from oracle.sttode_ref import first_diff_dup
from sttode_amd import packing, scenes
from sttode_amd.weights import make_weights


def _unpk32(tile):
    """inverse of packing.pk32_tile: 1024 floats -> [32 rows, 32 k]"""
    X = np.asarray(tile, np.float64).reshape(4, 2, 32, 4)       # g, h, i, r
    return X.transpose(2, 0, 1, 3).reshape(32, 32)             # i, (g, h, r) -> k = 8 g + 4 h + r


class _Feed:
    def __init__(self, pool, prog):
        self.seq = [int(f) + t for f, c in prog for t in range(int(c))]
        self.pool, self.i = pool, 0

    def next(self):
        self.i += 1
        return _unpk32(self.pool[self.seq[self.i - 1]])


def _emulate(rs, variant, Tp, x, last, xpad, g_in=None, attn=None, ode_time=12.0):
    """One pass of csrc/role32.hpp over all columns at once (float64): returns pf, state0, A0x, A0y, A1y."""
    C = packing.R32_CONSTS
    c = rs['consts_' + variant].astype(np.float64)
    fd = _Feed(rs['pool'], rs['prog_' + variant])
    n = xpad.shape[0]
    seg = lambda key, o, cnt: c[C[key] + o: C[key] + o + cnt]
    # block-0 conv + GRU (gate rows pre-scaled: sigmoid(a) = 1 / (1 + 2^a'), tanh(a) = 1 - 2 / (1 + 2^a'))
    d = np.zeros((n, 32)); d[:, :xpad.shape[1]] = xpad
    hs = np.zeros((n, 96))
    gb = c[C['gbias']: C['gbias'] + 384].reshape(4, 96)
    for t in range(Tp):
        e = np.maximum(seg('convb', 0, 32) + d @ fd.next().T, 0.0)
        hn = np.zeros_like(hs)
        for j in range(3):
            sl = slice(32 * j, 32 * j + 32)
            ar = gb[0, sl] + e @ fd.next().T
            for k in range(3):
                ar = ar + hs[:, 32 * k:32 * k + 32] @ fd.next().T
            r = 1.0 / (1.0 + np.exp2(ar))
            az = gb[1, sl] + e @ fd.next().T
            for k in range(3):
                az = az + hs[:, 32 * k:32 * k + 32] @ fd.next().T
            zg = 1.0 / (1.0 + np.exp2(az))
            an = gb[3, sl].copy()
            for k in range(3):
                an = an + hs[:, 32 * k:32 * k + 32] @ fd.next().T
            an = r * an + gb[2, sl]
            an = an + e @ fd.next().T
            ng = 1.0 - 2.0 / (1.0 + np.exp2(an))
            hn[:, sl] = zg * (hs[:, sl] - ng) + ng
        hs = hn
    # E
    if variant == 'scenes':
        G = np.tile(seg('bc', 0, 64), (n, 1)) + last[:, None] * seg('wlast', 0, 64)
        for j in range(2):
            for kt in range(rs['kte']):
                G[:, 32 * j:32 * j + 32] += x[:, 32 * kt:32 * kt + 32] @ fd.next().T
        src = G
    else:
        G, src = g_in.astype(np.float64), attn.astype(np.float64)
    XR = np.zeros((n, 64))
    for j in range(2):
        vi = seg('bi', 32 * j, 32) + src[:, :32] @ fd.next().T
        vi = vi + src[:, 32:] @ fd.next().T
        vg = seg('bg', 32 * j, 32) + src[:, :32] @ fd.next().T
        vg = vg + src[:, 32:] @ fd.next().T
        XR[:, 32 * j:32 * j + 32] = G[:, 32 * j:32 * j + 32] + np.tanh(vi) / (1.0 + np.exp(-vg))

    def ln(v, w, b):
        mu = v.mean(1, keepdims=True)
        var = ((v - mu) ** 2).mean(1, keepdims=True)
        return (v - mu) / np.sqrt(var + 1e-5) * seg(w, 0, 64) + seg(b, 0, 64)
    XR = ln(XR, 'ln1w', 'ln1b')
    FF = np.zeros((n, 64))
    for ht in range(32):
        hid = seg('l1b', 32 * ht, 32) + XR[:, :32] @ fd.next().T
        hid = np.maximum(hid + XR[:, 32:] @ fd.next().T, 0.0)
        FF[:, :32] += hid @ fd.next().T
        FF[:, 32:] += hid @ fd.next().T
    XR = ln(XR + (FF + seg('l2b', 0, 64)), 'ln2w', 'ln2b')
    Y = np.maximum(G + XR * ode_time, 0.0)
    pf = np.concatenate([G, Y], 1)
    B7 = np.concatenate([pf, hs], 1)                            # k-tiles 0..3 = pf, 4..6 = state0
    tabs = []
    for key, KT in (('b1x', 7), ('b1y', 7), ('b11', 4)):
        A = np.zeros((n, 512))
        for rt in range(16):
            acc = np.tile(seg(key, 32 * rt, 32), (n, 1))
            for kt in range(KT):
                acc = acc + B7[:, 32 * kt:32 * kt + 32] @ fd.next().T
            A[:, 32 * rt:32 * rt + 32] = acc
        tabs.append(A)
    assert fd.i == len(fd.seq), 'the program holds tiles the walk did not consume'
    return pf, hs, tabs


@pytest.mark.parametrize('case', ['eth', 'nba', 'nba_long'])
def test_role_stream_walk_matches_oracle(case):
    Tp, Tf = {'eth': (8, 12), 'nba': (5, 10), 'nba_long': (10, 40)}[case]
    sd = make_weights(1234, past_length=Tp, future_length=Tf)
    rs = packing.role_stream(sd, Tp)
    assert rs['prog_len_scenes'] == packing.role_prog_len(Tp, False) and rs['prog_len_nba'] == packing.role_prog_len(Tp, True)
    ora = oracle_model('eth' if case == 'eth' else 'nba', Tp, Tf)
    rng = np.random.default_rng(5)
    with torch.no_grad():
        if case == 'eth':
            obs, pred = scenes.eth_scene(77)
            ora.set_data(None, torch.from_numpy(obs), torch.from_numpy(pred))
            B, N = 1, obs.shape[0]
        else:
            B, N = 6, 10
            d = scenes.nba_batch(3, B, N=N, obs_len=Tp, pred_len=Tf)
            ora.set_data_nba({'past_traj': torch.from_numpy(d['past_traj']), 'future_traj': torch.from_numpy(d['future_traj'])})
        pt = ora.past_traj                                        # [n, Tp, 2] (normalised for scenes)
        inputs = torch.cat((pt, first_diff_dup(pt)), dim=-1)      # [n, Tp, 4]
        pf_ref, g_ref, _ = ora.past_encoder.trunk(inputs, B, N)
        blk0 = ora.decoder.decompose[0]
        e = torch.relu(blk0.conv_past(pt.transpose(1, 2))).transpose(1, 2)
        st_ref = blk0.encoder_past(e)[1].squeeze(0)
        feat = torch.cat((pf_ref, st_ref), 1)
        W = lambda b, nm: (ora.decoder.decompose[b].__getattr__('decoder_' + nm).layers[0].weight, ora.decoder.decompose[b].__getattr__('decoder_' + nm).layers[0].bias)
        tabs_ref = []
        for b, nm, with_state in ((0, 'x', True), (0, 'y', True), (1, 'y', False)):
            w, bb = W(b, nm)
            cols = torch.cat((w[:, :128], w[:, 160:256]), 1) if with_state else w[:, :128]
            tabs_ref.append((feat if with_state else pf_ref) @ cols.T + bb)
    n = pt.shape[0]
    x = np.zeros((n, 32 * rs['kte']))
    x[:, :4 * Tp] = inputs.numpy().reshape(n, 4 * Tp)
    xpad = pt.numpy().reshape(n, 2 * Tp)
    last = np.zeros(n)
    last[N - 1::N] = 1.0                                          # agent index N-1 of every scene (model/STTODE.py:206)
    if case == 'eth':
        pf, st, tabs = _emulate(rs, 'scenes', Tp, x, last, xpad)
    else:
        # attention groups > 1: g and the attention output (pre out_proj) come from the launches in front; here from the oracle's own layer
        layer = ora.past_encoder.ODE_Encoder.odeblock.odefunc.layers[0]
        mh = layer.self_attn.temporal_attention_before
        with torch.no_grad():
            src = g_ref.unsqueeze(2).reshape(B, N, 64)             # [L = B, Nb = N, 64]
            from oracle.sttode_ref import mhgsa
            eye, zero = torch.eye(64), torch.zeros(64)
            a_pre, _ = mhgsa(src, src, src, 8, mh.in_proj_weight, mh.in_proj_bias, eye, zero)   # identity out_proj: the pre-out_proj output
        pf, st, tabs = _emulate(rs, 'nba', Tp, x, last, xpad, g_in=g_ref.reshape(n, 64).numpy(), attn=a_pre.reshape(n, 64).numpy())
    tol = dict(rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(st, st_ref.numpy(), **tol)
    np.testing.assert_allclose(pf[:, :64], pf_ref.numpy()[:, :64], **tol)
    np.testing.assert_allclose(pf[:, 64:], pf_ref.numpy()[:, 64:], rtol=1e-4, atol=1e-4)     # x12-amplified FFN output, fp32 oracle
    for A, R in zip(tabs, tabs_ref):
        np.testing.assert_allclose(A, R.numpy(), rtol=1e-4, atol=1e-4)
